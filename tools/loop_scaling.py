#!/usr/bin/env python3
"""ns per sample of the carrier-loop kernels against the number of loops in one launch (one lane per loop, eight per wave): the
price of a loop should not depend on how many run beside it.  LS_N samples per loop; every loop its own input row (costas) or
eight loops per row (mpsk).  Run on the GPU box."""
import ctypes
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pymodem_amd  # noqa: E402
from pymodem_amd import taps as T  # noqa: E402
from pymodem_amd._native import Loop, check, lib  # noqa: E402

n = int(os.environ.get("LS_N", 1 << 20))
counts = [int(v) for v in os.environ.get("LS_LOOPS", "8,64,128,512,1024,2048").split(",")]
ctx = pymodem_amd.Context.default(0)
L = lib()
tab = ctx.upload(np.array([math.sin(i * 2.0 * math.pi / 256) for i in range(256)]))
pd = ctx.upload(np.ascontiguousarray(T.qpsk_error_table().reshape(-1), dtype=np.int32))
b0, b1, a1 = T.one_pole_lowpass(48000.0, 250.0, 1.0)
mx = max(counts)
rng = np.random.default_rng(3)
base = np.sin(np.arange(n) * 0.2) * 0.7
x1 = ctx.upload(np.concatenate([base * (1 + 0.01 * k) for k in range(min(mx, 64))] * (mx // min(mx, 64) + 1))[:n * mx])
x2 = ctx.upload(np.cos(np.arange(n) * 0.2) * 0.7)
o1 = ctx.empty(n * mx, np.float64)
o2 = ctx.empty(n * mx, np.float64)
for nl in counts:
    loops = (Loop * nl)()
    for k in range(nl):
        lp = loops[k]
        lp.phase_scaling, lp.index_scaling, lp.set_frequency = 2.0 * math.pi / 48000.0, 256 / (2.0 * math.pi), 1500.0 + 0.01 * k
        lp.b0, lp.b1, lp.a1 = b0, b1, a1
        lp.p_rate, lp.i_rate, lp.i_limit, lp.gain = 0.3, 0.3 / 2000, 31.25, 14400 / 65536
    res = {"loops": nl, "n": n}
    for name, fn in (("costas_own_rows", lambda: check(L.pm_costas_bpsk(ctx.handle, loops, nl, tab.ptr, x1.ptr, n, n, o1.ptr, n))),
                     ("costas_shared", lambda: check(L.pm_costas_bpsk(ctx.handle, loops, nl, tab.ptr, x1.ptr, 0, n, o1.ptr, n))),
                     ("mpsk_shared", lambda: check(L.pm_mpsk_loop(ctx.handle, loops, nl, tab.ptr, pd.ptr, x1.ptr, x2.ptr, 0, n, o1.ptr, o2.ptr, n)))):
        fn()
        ctx.timer_start()
        fn()
        res[name + "_ns"] = round(ctx.timer_stop() * 1e6 / n, 1)
    print(json.dumps(res), flush=True)
