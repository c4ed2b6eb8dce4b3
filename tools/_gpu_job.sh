cd $GRAFT_REPO_ROOT
timeout -k 5 60 ./tools/ubench/lat | head -8
timeout -k 10 600 python bench.py > gpurun_out/bench_also.json 2> gpurun_out/bench_also.err; echo rc=$?; tail -2 gpurun_out/bench_also.err; python - <<PY
import json
d=json.loads(open("gpurun_out/bench_also.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["h2d"], d["cpu_baseline"])
print(json.dumps(d["also"], indent=1)[:3000])
PY
