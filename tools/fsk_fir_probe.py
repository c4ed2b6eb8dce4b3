#!/usr/bin/env python3
"""The short-tap sign FIR of the fsk_9600 path (fir_short_signs_i16_kernel) alone on 2^28 samples, a few launches:
the target of the SQ / TCP / TCC counter passes of tools/collect_fsk_counters.sh.  Prints its HIP-event time."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pymodem_amd  # noqa: E402
from pymodem_amd._native import check, lib  # noqa: E402

NB = 1 << int(os.environ.get("PROBE_LOG2N", 28))
ctx = pymodem_amd.Context.default(0)
L = lib()
rng = np.random.default_rng(1)
xi = np.clip(np.rint(rng.standard_normal(1 << 20) * 8000), -32768, 32767).astype(np.int16)
big_in = ctx.upload(np.tile(xi, NB >> 20))
big_bits = ctx.empty(NB // 64 + 2, np.uint64)
h = ctx.upload(rng.standard_normal(8))
ctx.sync()
times = []
for _ in range(int(os.environ.get("PROBE_REPS", 6))):
    ctx.timer_start()
    check(L.pm_fir_signs_i16(ctx.handle, big_in.ptr, NB, h.ptr, 8, big_bits.ptr, 0))
    times.append(ctx.timer_stop())
print(json.dumps({"kernel": "fir_signs_i16 m=8", "n": NB, "ms": [round(t, 4) for t in times], "GB/s_best": round(2.125 * NB / min(times) / 1e6, 1)}))
