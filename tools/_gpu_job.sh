cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== bpsk chains/gpu=64"; python bench.py --workload bpsk_300 --chains-per-gpu 64 --samples 2880000 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['gpu_kernel_ms_per_step'], d['packets'])"
