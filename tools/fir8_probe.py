#!/usr/bin/env python3
"""The certified matched filter of the batch engine (fir8_kernel + fir8_exact_kernel, csrc/pm_fir8.hip) ALONE, at the engine's shapes:
rows x (chunk + m - 1) doubles -> sign bitmaps, for the two filters of the bench workloads (RRC 961: bpsk_300, RRC 241: qpsk_2400), next
to the exact binary64 rows kernel it replaces.  Run under `rocprofv3 --kernel-trace --stats` for the kernels' own durations (the entry
point used here makes its plan per call, which a HIP-event bracket would include)."""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pymodem_amd  # noqa: E402
from pymodem_amd import chain_builder as cb  # noqa: E402
from pymodem_amd._native import check, lib  # noqa: E402

ctx = pymodem_amd.Context.default(0)
L = lib()
rng = np.random.default_rng(1)
out = []
for kind, cfg, rows in (("bpsk", "300", int(os.environ.get("PROBE_ROWS_BPSK", 2048))), ("mpsk", "qpsk_2400", int(os.environ.get("PROBE_ROWS_QPSK", 4096)))):
    md = cb.ModemConfigurator(48000, {"type": kind, "config": cfg, "options": {}})
    taps = np.asarray(md.rrc_taps, dtype=np.float64)
    m = len(taps)
    chunk = 65536
    n = chunk + m - 1
    pitch = (n + 7) // 8 * 8
    base = (rng.standard_normal(1 << 22) * 0.35).clip(-1.2, 1.2)
    x = ctx.upload(np.resize(base, rows * pitch))
    stride = chunk // 64
    bits = ctx.empty(rows * stride, np.uint64)
    dt = ctx.upload(taps)
    redo = ctypes.c_int64()
    ctx.sync()
    t8, tx = [], []
    for _ in range(int(os.environ.get("PROBE_REPS", 4))):
        ctx.timer_start()
        check(L.pm_fir8_rows_signs_f64(ctx.handle, x.ptr, pitch, rows, n, taps.ctypes.data_as(ctypes.c_void_p), m, bits.ptr, stride, ctypes.byref(redo)))
        t8.append(ctx.timer_stop())
    if not os.environ.get("PROBE_NO_EXACT"):
        for _ in range(2):
            ctx.timer_start()
            check(L.pm_fir_rows_signs_f64(ctx.handle, x.ptr, pitch, rows, n, dt.ptr, m, bits.ptr, stride, 0))
            tx.append(ctx.timer_stop())
    outs = rows * chunk
    out.append({"filter": f"{kind} RRC {m}", "rows": rows, "chunk": chunk, "outputs": outs, "recomputed_exactly": redo.value,
                "fir8_call_ms_incl_plan": [round(t, 3) for t in t8], "exact_rows_kernel_ms": [round(t, 3) for t in tx],
                "fir8_Goutputs_per_s_best_call": round(outs / min(t8) / 1e6, 1), "exact_Goutputs_per_s": round(outs / min(tx) / 1e6, 1) if tx else None,
                "input_GB": round(rows * n * 8 / 1e9, 3)})
print(json.dumps(out))
