cd $GRAFT_REPO_ROOT
for OV in 0 1; do
echo "== signal overlap=$OV"; python bench.py --steps 8 --warmup 2 --no-cpu-baseline --overlap $OV 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], sum(d['gpu_kernel_ms_per_step'].values()), d['packets'])"
echo "== noise overlap=$OV"; python bench.py --steps 8 --warmup 2 --buffer noise --no-cpu-baseline --overlap $OV 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], sum(d['gpu_kernel_ms_per_step'].values()), d['packets'])"
echo "== fsk overlap=$OV"; python bench.py --workload fsk_9600 --steps 8 --warmup 2 --no-cpu-baseline --overlap $OV 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], sum(d['gpu_kernel_ms_per_step'].values()), d['packets'])"
done
echo "== 2 ranks gloo"; python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 --backend gloo --samples 9600000 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['ms_per_step'], d['config']['chains_total'], d['packets'])"
