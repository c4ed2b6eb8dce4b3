cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 &&
for i in 1 2 3; do timeout -k 10 200 python bench.py --no-cpu-baseline --also 0 > gpurun_out/p.json 2> gpurun_out/p.err; python - <<PY
import json
d=json.loads(open("gpurun_out/p.json").read().strip().splitlines()[-1])
print("run",$i,d["value"],d["ms_per_step"],d["gpu_kernel_ms_per_step"],d["pipeline_stage_ms_per_step"])
PY
done
timeout -k 10 200 python bench.py --no-cpu-baseline --also 0 --overlap 0 --steps 10 > gpurun_out/p.json 2> gpurun_out/p.err; python - <<PY
import json
d=json.loads(open("gpurun_out/p.json").read().strip().splitlines()[-1])
print("overlap0",d["value"],d["ms_per_step"],d["gpu_kernel_ms_per_step"])
PY
