#!/usr/bin/env python3
"""The certified AFSK sweep (pm_afsk_sweep_signs_tones, 7 gains and 1 gain) alone on the bench recording's band-passed stream, a few
launches: the target of counter passes (tools/_scratch / collect scripts).  PM_AFSK_LPF64=1 selects the kernel with binary64 low-passes."""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import pymodem_amd  # noqa: E402
from pymodem_amd import chain_builder as cb  # noqa: E402
from pymodem_amd._native import check, lib  # noqa: E402
from pymodem_amd.modems import AFSKModem  # noqa: E402


class A:
    pass


a = A()
a.samples, a.rate, a.workload, a.buffer = int(os.environ.get("SP_N", 28_800_000)), 48000, "afsk_1200_super_opt", "signal"
audio = bench.make_buffer(a)
ctx = pymodem_amd.Context.default(0)
d_audio = ctx.upload(audio)
lines = [bench.wl_afsk_super_opt(c) for c in range(8)]
modems = [cb.ModemConfigurator(48000, ln["modem"]) for ln in lines]
for md in modems:
    md.use_context(ctx)
bpf = modems[1].front_end(d_audio)
bound = float(np.abs(modems[1].input_bpf).sum()) * 32768.0
ctx.sync()
for name, mods in (("g=7", modems[1:]), ("g=1", modems[:1])):
    times = []
    for _ in range(int(os.environ.get("SP_REPS", 5))):
        ctx.timer_start()
        AFSKModem.sweep_signs(mods, bpf, bound)
        times.append(ctx.timer_stop())
    ctx.profile(True)                                        # the library's own per-class HIP-event timing: fused kernel / exact kernel apart
    for _ in range(5):
        AFSKModem.sweep_signs(mods, bpf, bound)
    ctx.sync()
    prof = {k: round(v[0] / max(v[1], 1), 4) for k, v in ctx.profile_read().items() if v[1]}
    ctx.profile(False)
    print(json.dumps({"sweep": name, "ms": [round(t, 4) for t in times], "uncertain": AFSKModem.sweep_uncertain(ctx), "avg_ms_by_class": prof}))
