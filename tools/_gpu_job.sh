cd $GRAFT_REPO_ROOT
for i in 1 2; do PYMODEM_AMD_FORCE_GATHER=1 timeout -k 10 200 python bench.py --no-cpu-baseline --also 0 > gpurun_out/force_nccl.json 2> gpurun_out/force_nccl.err; echo rc=$?; python - <<PY
import json
d=json.loads(open("gpurun_out/force_nccl.json").read().strip().splitlines()[-1])
print("nccl 1-rank forced gather", d["value"], d["ms_per_step"], d["pipeline_stage_ms_per_step"], d["packets"])
PY
done
