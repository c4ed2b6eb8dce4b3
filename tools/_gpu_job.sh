set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "slicer" 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 2>/dev/null > gpurun_out/bench_afsk.json
python -c "import json; d=json.load(open('gpurun_out/bench_afsk.json')); print(d['value'], d['ms_per_step'], d['gpu_kernel_ms_per_step'], d['cpu_baseline'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_write.log 2>&1
ls -la $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch/* $GRAFT_REPO_ROOT/gpurun_out/pmc_write/* | head
