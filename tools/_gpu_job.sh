cd $GRAFT_REPO_ROOT
for w in "afsk_1200_super_opt" "fsk_9600" "afsk_1200_super_opt"; do timeout -k 10 200 python bench.py --no-cpu-baseline --also 0 --workload $w > gpurun_out/p.json 2> gpurun_out/p.err; python - <<PY
import json
d=json.loads(open("gpurun_out/p.json").read().strip().splitlines()[-1])
print("$w",d["value"],d["ms_per_step"],d["pipeline_stage_ms_per_step"])
PY
done
