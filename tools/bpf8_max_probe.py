#!/usr/bin/env python3
"""max(band-passed recording) on the matrix pipe (pm_bpf8.hip: bpf8_max_kernel) alone: ROWS rows of N samples of the bpsk_300 bench
buffer through pm_bpf8_rows_max_i16, a few calls -- the target of `rocprofv3 --kernel-trace --stats` (the entry plans its tables and
allocates per call, so the wall time printed here is an upper bound; the kernel's own time is in the trace)."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import pymodem_amd  # noqa: E402
from pymodem_amd import chain_builder as cb  # noqa: E402
from pymodem_amd._native import check, lib  # noqa: E402

ROWS, N = int(os.environ.get("SP_ROWS", 64)), int(os.environ.get("SP_N", 7_200_000))


class A:
    pass


a = A()
a.samples, a.rate, a.workload, a.buffer = ROWS * N, 48000, "bpsk_300", "signal"
audio = bench.make_buffer(a)
ctx = pymodem_amd.Context.default(0)
d_audio = ctx.upload(audio)
md = cb.ModemConfigurator(48000, bench.WORKLOADS["bpsk_300"][0](0)["modem"])
h = np.ascontiguousarray(md.input_bpf, np.float64)
out = np.zeros(ROWS)
redone = ctypes.c_int64()
for _ in range(int(os.environ.get("SP_REPS", 5))):
    t0 = time.perf_counter()
    check(lib().pm_bpf8_rows_max_i16(ctx.handle, d_audio.ptr, N, ROWS, N, h.ctypes.data, len(h), out.ctypes.data, ctypes.byref(redone)))
    dt = time.perf_counter() - t0
    print("taps %d rows %d x %d samples: %.3f ms per call, %.4f ms per 28.8 M samples; %d outputs through the exact chain (%.1f per row)"
          % (len(h), ROWS, N, dt * 1e3, dt * 1e3 / (ROWS * N / 28.8e6), redone.value, redone.value / ROWS))
if os.environ.get("SP_CHECK", "1") == "1":       # against the bit-exact FIR kernel's own maximum (tests/ pin that kernel to the oracle)
    dh = ctx.upload(h)
    y = ctx.empty(N - len(h) + 1, np.float64)
    want = []
    for r in range(min(ROWS, 4)):
        check(lib().pm_fir_valid_i16(ctx.handle, d_audio.view(r * N, N).ptr, N, dh.ptr, len(h), y.ptr, 0))
        want.append(y.download().max())
    print("first rows equal the maximum of pm_fir_valid_i16:", out[:len(want)].tobytes() == np.array(want).tobytes())
