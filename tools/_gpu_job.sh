cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
echo "== bpsk full (signal)"; PM_AGC_TRACE=1 python bench.py --workload bpsk_300 --steps 1 --warmup 1 --no-cpu-baseline 2>gpurun_out/agc_trace.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['gpu_kernel_ms_per_step'], d['packets'])"; tail -3 gpurun_out/agc_trace.txt
echo "== bpsk full (noise)"; PM_AGC_TRACE=1 python bench.py --workload bpsk_300 --steps 1 --warmup 1 --buffer noise --no-cpu-baseline 2>gpurun_out/agc_trace2.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['gpu_kernel_ms_per_step'], d['packets'])"; tail -2 gpurun_out/agc_trace2.txt
echo "== qpsk 8 chains full"; python bench.py --workload qpsk_2400 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['gpu_kernel_ms_per_step'], d['packets'])"
