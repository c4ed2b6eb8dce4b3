cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
echo "== signal"; python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['gpu_kernel_ms_per_step'], d['packets'])"
echo "== noise"; python bench.py --steps 5 --warmup 2 --buffer noise --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['gpu_kernel_ms_per_step'], d['packets'])"
echo "== fsk"; python bench.py --workload fsk_9600 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['gpu_kernel_ms_per_step'], d['packets'])"
