cd $GRAFT_REPO_ROOT
for sw in 3 2; do for cw in 110 160 220 330; do
  export PM_SLICER_CHUNK_WORDS=$cw
  timeout -k 10 200 python bench.py --steps 24 --warmup 4 --no-cpu-baseline --slice-workers $sw > gpurun_out/sweep.json 2> gpurun_out/sweep.err; python - <<PY
import json
d=json.loads(open("gpurun_out/sweep.json").read().strip().splitlines()[-1])
print("workers",$sw,"chunkwords",$cw,d["value"],d["ms_per_step"],d["pipeline_stage_ms_per_step"], d["slicer"], d["gpu_kernel_ms_per_step"])
PY
done; done
