cd $GRAFT_REPO_ROOT
for mode in none gloo nccl; do
  if [ $mode = none ]; then timeout -k 10 200 python bench.py --steps 16 --warmup 3 --no-cpu-baseline > gpurun_out/m_$mode.json 2> gpurun_out/m_$mode.err
  else PYMODEM_AMD_FORCE_GATHER=1 timeout -k 10 200 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --backend $mode > gpurun_out/m_$mode.json 2> gpurun_out/m_$mode.err; fi
  python - <<PY
import json
d=json.loads(open("gpurun_out/m_$mode.json").read().strip().splitlines()[-1])
print("$mode", d["value"], d["ms_per_step"], d["pipeline_stage_ms_per_step"], d["gpu_kernel_ms_per_step"])
PY
done
