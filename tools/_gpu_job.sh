cd $GRAFT_REPO_ROOT
echo "== N=1 signal buffer"; python bench.py --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['gpu_kernel_ms_per_step'], d['slicer'], d['packets'], d['cpu_baseline'])"
echo "== N=1 noise buffer"; python bench.py --steps 5 --warmup 2 --buffer noise --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['gpu_kernel_ms_per_step'], d['slicer'], d['packets'])"
echo "== fsk"; python bench.py --workload fsk_9600 --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['gpu_kernel_ms_per_step'], d['slicer'], d['packets'], d['cpu_baseline'])"
python - <<'PY'
import cProfile, pstats, sys, json, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.argv=["bench.py"]
import numpy as np, bench, pymodem_amd
from pymodem_amd import chain_builder as cb, chain_execute as ce, dist as pdist
class A: samples=28_800_000; rate=48000; buffer='signal'; workload='afsk_1200_super_opt'
ctx=pymodem_amd.Context.default(0); d_audio=ctx.upload(bench.make_buffer(A))
lines=[bench.wl_afsk_super_opt(c) for c in range(8)]; modems=[cb.ModemConfigurator(48000,l["modem"]) for l in lines]
def step():
    chains=[[l["object_name"], m, cb.SlicerConfigurator(m.output_sample_rate,l["slicer"]), cb.StreamConfigurator(l["stream"]), cb.CodecConfigurator(l["codec"],l["object_name"])] for l,m in zip(lines,modems)]
    pk=dict(enumerate(ce.process_chains_device(chains,d_audio)))
    return pdist.correlate(pdist.gather_packets(pk,[l["object_name"] for l in lines]),8,1200)
step(); step()
cProfile.run("step()","/tmp/s.prof"); pstats.Stats("/tmp/s.prof").sort_stats("cumulative").print_stats(22)
PY
