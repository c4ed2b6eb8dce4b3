#!/usr/bin/env python3
"""The group band-pass alone on the bench recording: fir_valid_kernel<short> (binary64, the reference's sum) and bpf8_kernel (int8
matrix pipe, value with a bound), a few launches each -- the target of `rocprofv3 --kernel-trace --stats`."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import pymodem_amd  # noqa: E402
from pymodem_amd import chain_builder as cb  # noqa: E402
from pymodem_amd._native import check, lib  # noqa: E402


class A:
    pass


a = A()
a.samples, a.rate, a.workload, a.buffer = int(os.environ.get("SP_N", 28_800_000)), 48000, "afsk_1200_super_opt", "signal"
audio = bench.make_buffer(a)
ctx = pymodem_amd.Context.default(0)
d_audio = ctx.upload(audio)
md = cb.ModemConfigurator(48000, bench.wl_afsk_super_opt(1)["modem"])
h = np.ascontiguousarray(md.input_bpf, np.float64)
dh = ctx.upload(h)
n = len(audio)
y0 = ctx.empty(n - len(h) + 1, np.float64)
y1 = ctx.empty(n - len(h) + 1, np.float64)
bound = ctypes.c_double()
for _ in range(int(os.environ.get("SP_REPS", 6))):
    ctx.timer_start()
    check(lib().pm_fir_valid_i16(ctx.handle, d_audio.ptr, n, dh.ptr, len(h), y0.ptr, 0))
    t0 = ctx.timer_stop()
    check(lib().pm_fir_valid_i16_limbs(ctx.handle, d_audio.ptr, n, h.ctypes.data, len(h), y1.ptr, ctypes.byref(bound)))
    print("f64 %.4f ms" % t0)
a0, a1 = y0.download(), y1.download()
print("taps", len(h), "max |limbs - f64|", float(np.abs(a0 - a1).max()), "bound", bound.value, "sum|h|*32768", float(np.abs(h).sum() * 32768))
