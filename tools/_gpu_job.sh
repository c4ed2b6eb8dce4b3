cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_qpsk_modem.py tests/test_segments.py tests/test_gpu_chains.py -m gpu -x -q 2>&1 | tail -2 &&
KB_BIG=0 KB_REPS=2 timeout -k 10 600 python tools/kernel_bench.py 2>&1 | grep "costas\|mpsk\|agc"
