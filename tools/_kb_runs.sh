for r in 16 10 12 34 16; do
  PM_FUSE_RUN=$r KB_ONLY=fir KB_BIG=0 python tools/kernel_bench.py 2>/dev/null | grep "afsk_sweep_signs_tones.*fused" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('run $r', d['kernel'][:32], d['ms_best'], d['ms_avg'])"
done
