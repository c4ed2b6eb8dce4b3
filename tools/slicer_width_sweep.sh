#!/bin/bash
# Run on the GPU box: the native executor's slicer batching -- recordings per slicer batch (slice_group), bitmap
# slots, slicer workers -- on the final demod kernels, 400 steps and the driver's 20, same box, interleaved twice.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/slicer_sweep; mkdir -p $OUT
run() { # label steps warm env...
  label=$1; steps=$2; warm=$3; shift 3
  env "$@" timeout -k 10 200 python3 bench.py --gpus 1 --steps $steps --warmup $warm --also 0 --no-cpu-baseline --with-exchange 0 > $OUT/$label.json 2> $OUT/$label.err || echo "$label failed"
  python3 -c "
import json
d=json.loads(open('$OUT/$label.json').read().strip().splitlines()[-1])
k=d['gpu_kernel_ms_per_step']; s=d.get('pipeline_stage_ms_per_step') or {}
print('%-34s steps %4d  %.3f ms/step  %7.1f Gsamples/s  steady %s  slice_iter %.3f slice_emit %.3f  rec/batch %s' % ('$label', $steps, d['ms_per_step'], d['value']/1e3, d.get('steady_state_ms_per_step'), k.get('slice_iter',0), k.get('slice_emit',0), s.get('recordings_per_slice_batch')))"
}
for rep in 1 2; do
for cfg in "g4_s16_w2 PYMODEM_AMD_PIPE_GROUP=4 PYMODEM_AMD_PIPE_SLOTS=16 PYMODEM_AMD_PIPE_WORKERS=2" \
           "g8_s16_w2 PYMODEM_AMD_PIPE_GROUP=8 PYMODEM_AMD_PIPE_SLOTS=16 PYMODEM_AMD_PIPE_WORKERS=2" \
           "g8_s32_w2 PYMODEM_AMD_PIPE_GROUP=8 PYMODEM_AMD_PIPE_SLOTS=32 PYMODEM_AMD_PIPE_WORKERS=2" \
           "g16_s32_w2 PYMODEM_AMD_PIPE_GROUP=16 PYMODEM_AMD_PIPE_SLOTS=32 PYMODEM_AMD_PIPE_WORKERS=2" \
           "g8_s32_w1 PYMODEM_AMD_PIPE_GROUP=8 PYMODEM_AMD_PIPE_SLOTS=32 PYMODEM_AMD_PIPE_WORKERS=1" \
           "g16_s32_w1 PYMODEM_AMD_PIPE_GROUP=16 PYMODEM_AMD_PIPE_SLOTS=32 PYMODEM_AMD_PIPE_WORKERS=1" \
           "g4_s32_w2 PYMODEM_AMD_PIPE_GROUP=4 PYMODEM_AMD_PIPE_SLOTS=32 PYMODEM_AMD_PIPE_WORKERS=2" \
           "g2_s16_w2 PYMODEM_AMD_PIPE_GROUP=2 PYMODEM_AMD_PIPE_SLOTS=16 PYMODEM_AMD_PIPE_WORKERS=2"; do
  set -- $cfg; name=$1; shift
  run ${name}_400_r$rep 400 10 "$@"
  run ${name}_20_r$rep 20 5 "$@"
done
done
