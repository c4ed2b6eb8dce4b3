cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -2 &&
KB_ONLY=fir KB_REPS=5 timeout -k 10 300 python tools/kernel_bench.py 2>&1 | grep "signs"
