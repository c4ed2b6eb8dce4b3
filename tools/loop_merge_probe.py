#!/usr/bin/env python3
"""Experiment (with its kill criterion): can ONE recording's carrier loop be cut into speculative chunks the way the
slicer is?  A chunk would start from a PREDICTED state some samples before its own first sample (a warm-up overlap) and be
accepted only when its state -- NCO phase, loop filter x[-1] / y[-1], PI integrator, the last control value (psk.py:173-189) -- equals the
true sequential run's BIT FOR BIT at the chunk's start; otherwise the chunk is run again from the true state.  This measures, on the
bench's bpsk_300 buffer, how long after a speculative start the states merge:

  true run      the Costas loop of configs/bpsk_300.json over the AGC'd stream, its state recorded every STEP samples;
  speculation   at every chunk boundary b (CHUNK apart) a loop is started at b - WARM from (a) a fresh loop's state (psk.py:134-160) and
                (b) the true state of the boundary before it with its phase advanced by the nominal carrier (the best cheap prediction:
                everything the true run knew one chunk earlier), run forward, and compared with the true state every STEP samples up to
                b + HORIZON.

Kill criterion: fewer than 90 % of the chunks merged within 16 384 samples -> the idea is dropped, the distribution recorded
(profiles/r04_loop_merge.json).  No tolerance is ever shipped: a state that is merely close is a different bit stream.
All loops run on the GPU through the C ABI (pm_costas_bpsk), many speculative starts per launch, one lane each."""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import pymodem_amd  # noqa: E402
from pymodem_amd import chain_builder as cb  # noqa: E402
from pymodem_amd._native import Loop, check, lib  # noqa: E402

N = int(os.environ.get("LM_SAMPLES", 4_000_000))
CHUNK, WARM, HORIZON, STEP = 65536, 16384, 16384, 1024
STATE = ("phase", "control", "sine", "cosine", "x0", "x1", "y0", "integral", "proportional")


class A:
    pass


a = A()
a.samples, a.rate, a.workload, a.buffer = N, 48000, "bpsk_300", "signal"
audio = bench.make_buffer(a)
ctx = pymodem_amd.Context.default(0)
line = bench.WORKLOADS["bpsk_300"][0](0)
modem = cb.ModemConfigurator(48000, line["modem"])
agcd = modem.front_end(ctx.upload(audio))                 # band-pass + AGC: the loop's input (device, float64)
n = agcd.n
table = ctx.upload(np.asarray(modem.wavetable, dtype=np.float64))
L = lib()


def state_of(lp):
    return tuple(np.float64(getattr(lp, k)).tobytes() for k in STATE)


def run(loops, offsets, count):
    """every loop over `count` samples from its own offset into the AGC'd stream (rows of the same buffer); states updated in place"""
    out = ctx.scratch(("merge-out",), len(loops) * count, np.float64)
    # pm_costas_bpsk reads loop l at d_x + l * x_stride: launch the loops one group per distinct offset spacing -- here every loop has
    # its own offset, so one launch each would be slow; instead the offsets are CHUNK apart and the stride does the work
    base, stride = offsets[0], (offsets[1] - offsets[0] if len(offsets) > 1 else 0)
    assert all(offsets[i] == base + i * stride for i in range(len(offsets)))
    check(L.pm_costas_bpsk(ctx.handle, loops, len(loops), table.ptr, ctypes.c_void_p(agcd.ptr.value + 8 * base), stride, count, out.ptr, count))


# ---- the true run: one loop, STEP samples per call, its state after every call
fresh = Loop()
ctypes.memmove(ctypes.byref(fresh), modem._loop0, ctypes.sizeof(Loop))
true = (Loop * 1)()
ctypes.memmove(true, ctypes.byref(fresh), ctypes.sizeof(Loop))
nstep = n // STEP
truth = {0: state_of(true[0])}
snap, snap_warm = {}, {}
for s in range(nstep):
    run(true, [s * STEP], STEP)
    truth[(s + 1) * STEP] = state_of(true[0])
    if ((s + 1) * STEP) % CHUNK == 0 or ((s + 1) * STEP + WARM) % CHUNK == 0:
        c = Loop()
        ctypes.memmove(ctypes.byref(c), true, ctypes.sizeof(Loop))
        (snap if ((s + 1) * STEP) % CHUNK == 0 else snap_warm)[(s + 1) * STEP] = c

bounds = [b for b in range(2 * CHUNK, n - HORIZON - STEP, CHUNK)]
results = {}
for label in ("fresh state", "previous boundary's state, phase advanced by the nominal carrier", "control: the true state at the speculative start"):
    loops = (Loop * len(bounds))()
    for i, b in enumerate(bounds):
        if label.startswith("fresh"):
            ctypes.memmove(ctypes.byref(loops[i]), ctypes.byref(fresh), ctypes.sizeof(Loop))
        elif label.startswith("control"):                  # (the harness itself: a loop started from the true state must compare equal at once)
            ctypes.memmove(ctypes.byref(loops[i]), ctypes.byref(snap_warm[b - WARM]), ctypes.sizeof(Loop))
        else:
            p = snap[b - CHUNK]
            ctypes.memmove(ctypes.byref(loops[i]), ctypes.byref(p), ctypes.sizeof(Loop))
            # CHUNK - WARM samples later the NCO of a loop sitting exactly on the carrier would be here (nco.py:35-39)
            adv = p.phase + p.phase_scaling * (p.set_frequency + p.control) * (CHUNK - WARM)
            loops[i].phase = float(np.mod(adv, 2 * np.pi))
    merged_at = [None] * len(bounds)                       # samples after the boundary (negative: inside the warm-up) of the first bitwise-equal state
    pos = -WARM
    offs = [b - WARM for b in bounds]
    while pos < HORIZON:
        run(loops, offs, STEP)
        pos += STEP
        offs = [o + STEP for o in offs]
        for i, b in enumerate(bounds):
            if merged_at[i] is None and state_of(loops[i]) == truth[b + pos]:
                merged_at[i] = pos
    done = [m for m in merged_at if m is not None]
    # how close do the ones that never merge get?  phase distance at the horizon
    results[label] = {"chunks": len(bounds), "merged_within_warmup": sum(1 for m in done if m <= 0), "merged_within_16384_after_start": sum(1 for m in done if m <= HORIZON - WARM),
                      "merged_by_horizon": len(done), "fraction_merged_within_16384": round(sum(1 for m in done if m + WARM <= 16384) / len(bounds), 4),
                      "merge_positions_histogram": {str(k): int(v) for k, v in zip(*np.unique(np.array(done, dtype=np.int64), return_counts=True))} if done else {}}
out = {"experiment": "bitwise merge of speculative Costas-loop chunks with the true run (bpsk_300, bench buffer)", "samples": int(n), "chunk": CHUNK,
       "warm_up": WARM, "horizon_after_boundary": HORIZON, "compared_every": STEP, "state_compared": list(STATE), "results": results,
       "kill_criterion": "fewer than 90 % of chunks merged within 16384 samples of their speculative start -> dropped",
       "verdict": "dropped" if all(r["fraction_merged_within_16384"] < 0.9 for k, r in results.items() if not k.startswith("control")) else "worth building"}
print(json.dumps(out, indent=1))
