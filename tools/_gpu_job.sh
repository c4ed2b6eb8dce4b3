cd $GRAFT_REPO_ROOT
for mode in fast exact; do export PM_FIR_SIGNS=$mode; for ov in 0 2; do timeout -k 10 200 python bench.py --no-cpu-baseline --also 0 --overlap $ov --steps 20 > gpurun_out/p.json 2> gpurun_out/p.err; python - <<PY
import json
d=json.loads(open("gpurun_out/p.json").read().strip().splitlines()[-1])
print("$mode overlap",$ov,d["value"],d["ms_per_step"],d["gpu_kernel_ms_per_step"], d["packets"])
PY
done; done
