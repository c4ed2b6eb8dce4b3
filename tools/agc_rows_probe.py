#!/usr/bin/env python3
"""agc_rows_kernel alone: R rows of n samples (pm_agc_rows_apply), ms per call and ns per sample and row step.  Run on the GPU box."""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pymodem_amd  # noqa: E402
from pymodem_amd._native import AGCParams, check, lib  # noqa: E402

n = int(os.environ.get("AG_N", 131072))
ctx = pymodem_amd.Context.default(0)
for rows in [int(v) for v in os.environ.get("AG_ROWS", "16,512,2048,8192").split(",")]:
    rng = np.random.default_rng(1)
    x = ctx.upload((rng.standard_normal(n) * 3000.0).repeat(1))
    xs = ctx.empty(rows * n, np.float64)
    for r in range(0, rows, max(rows // 8, 1)):
        pass
    # the same row everywhere is fine for timing (the step is branch-free)
    check(lib().pm_memset(ctx.handle, xs.ptr, 0, rows * n * 8))
    for r in range(rows):
        if r < 64 or r % 97 == 0:
            check(lib().pm_d2d(ctx.handle, xs.ptr.value + r * n * 8, x.ptr, n * 8))
    y = ctx.empty(rows * n, np.float64)
    p = AGCParams(500.0, 50.0, 0.00025, 48000.0, 10000.0)
    normal = (ctypes.c_double * rows)(*([12000.0] * rows))
    state = (ctypes.c_double * (2 * rows))()
    ms = []
    for _ in range(3):
        ctx.timer_start()
        check(lib().pm_agc_rows_apply(ctx.handle, xs.ptr, n, y.ptr, n, rows, n, ctypes.byref(p), normal, state))
        ms.append(ctx.timer_stop())
    print(json.dumps({"rows": rows, "n": n, "ms": [round(v, 3) for v in ms], "ns_per_sample_step": round(min(ms) * 1e6 / n, 1)}), flush=True)
