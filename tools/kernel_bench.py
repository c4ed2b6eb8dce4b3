#!/usr/bin/env python3
"""Per-kernel timing through the C ABI (HIP events on the ctx stream): GB/s against algorithmic bytes and
GFMA/s against the f64 vector peak, for the tap counts the BASELINE configs use.  Not a test; run on the GPU box."""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pymodem_amd  # noqa: E402
from pymodem_amd._native import AGCParams, Loop, SlicerParams, check, lib  # noqa: E402

N = int(os.environ.get("KB_N", 28_800_000))
REPS = int(os.environ.get("KB_REPS", 5))
ctx = pymodem_amd.Context.default(0)
L = lib()
rng = np.random.default_rng(1)
xi = np.clip(np.rint(rng.standard_normal(N) * 8000), -32768, 32767).astype(np.int16)
d_i16 = ctx.upload(xi)
d_f64 = ctx.upload(xi.astype(np.float64) * 0.01)
d_out = ctx.empty(N, np.float64)
d_out2 = ctx.empty(N, np.float64)
rows = []


def timeit(fn, reps=REPS):
    fn()
    ctx.sync()
    best = 1e30
    tot = 0.0
    for _ in range(reps):
        ctx.timer_start()
        fn()
        ms = ctx.timer_stop()
        best = min(best, ms)
        tot += ms
    return best, tot / reps


def report(name, ms, bytes_alg, fma=0):
    row = {"kernel": name, "ms_best": round(ms[0], 4), "ms_avg": round(ms[1], 4), "GB/s": round(bytes_alg / ms[0] / 1e6, 1),
           "frac_hbm_8TBs": round(bytes_alg / ms[0] / 1e6 / 8000, 4)}
    if fma:
        row["TFMA/s"] = round(fma / ms[0] / 1e9, 3)
        row["frac_f64_fma_peak"] = round(fma / ms[0] / 1e9 / 39.3, 4)
    rows.append(row)
    print(json.dumps(row), flush=True)


for m in [8, 40, 100, 148, 240, 961]:
    h = ctx.upload(rng.standard_normal(m))
    report(f"fir_i16 m={m}", timeit(lambda: check(L.pm_fir_valid_i16(ctx.handle, d_i16.ptr, N, h.ptr, m, d_out.ptr, 0))), 10.0 * N, m * N)
for m in [8, 100, 163, 241, 961]:
    h = ctx.upload(rng.standard_normal(m))
    report(f"fir_f64 m={m}", timeit(lambda: check(L.pm_fir_valid_f64(ctx.handle, d_f64.ptr, N, h.ptr, m, d_out.ptr, 0))), 16.0 * N, m * N)
for m in [40, 60]:
    t = [ctx.upload(rng.standard_normal(m)) for _ in range(4)]
    report(f"afsk_correlate m={m}", timeit(lambda: check(L.pm_afsk_correlate(ctx.handle, d_f64.ptr, N, t[0].ptr, t[1].ptr, t[2].ptr, t[3].ptr, m, d_out.ptr))),
           16.0 * N, 4 * m * N)
bits = ctx.empty(N // 64 + 2, np.uint64)
for m in [100, 241]:
    h = ctx.upload(rng.standard_normal(m))
    report(f"fir_signs_f64 m={m}", timeit(lambda: check(L.pm_fir_signs_f64(ctx.handle, d_f64.ptr, N, h.ptr, m, bits.ptr, 0))), 8.125 * N, m * N)
for g, m in [(7, 60), (4, 40)]:
    t = [ctx.upload(rng.standard_normal(m)) for _ in range(2)]
    sp = ctx.upload(rng.standard_normal(g * 2 * m))
    stride = (N - m + 1 + 63) // 64 * 64
    og = ctx.scratch("kb_group", stride * g, np.float64)
    report(f"afsk_correlate_group g={g} m={m}", timeit(lambda: check(L.pm_afsk_correlate_group(ctx.handle, d_f64.ptr, N, t[0].ptr, t[1].ptr, sp.ptr, g, m, og.ptr, stride))),
           8.0 * N * (1 + g), (2 + 2 * g) * m * N)
report("signs", timeit(lambda: check(L.pm_signs_f64(ctx.handle, d_f64.ptr, N, bits.ptr))), 8.125 * N)

# the certified AFSK path on the headline templates: magnitudes by direct and by sliding sums, then the whole sweep
# entry (7 modems / 1 modem) fused and as separate kernels
from pymodem_amd import taps as TT  # noqa: E402
from pymodem_amd._native import AfskTones  # noqa: E402
mi, mq, ui, uq = TT.afsk_tone_correlators(48000.0, 1200.0, 1300.0, 2100.0, 1.0, 1.5, 0.0)
mc = len(mi)
mk, sp = TT.tone_model(mi, mq), TT.tone_model(ui, uq)
tones = AfskTones()
tones.mark_rot[:], tones.mark_end[:], tones.space_rot[:], tones.space_end[:], tones.tap_dev = mk[0], mk[1], sp[0], sp[1], max(mk[2], sp[2])
dt = [ctx.upload(v) for v in (mi, mq, ui, uq)]
xb = float(np.abs(xi).max()) * 0.01
for name, tp, fma in (("direct sums", None, 4 * mc), ("sliding sums", ctypes.byref(tones), 4 * mc / 16 + 9)):
    report(f"afsk_magnitudes {name} m={mc}", timeit(lambda: check(L.pm_afsk_magnitudes(ctx.handle, d_f64.ptr, N, xb, dt[0].ptr, dt[1].ptr, dt[2].ptr,
                                                                                     dt[3].ptr, mc, tp, d_out.ptr, d_out2.ptr, None))), 24.0 * N, fma * N)
lpf = TT.windowed_sinc(100, 900.0, 48000.0, pass_zero=True)
dl = ctx.upload(lpf)
for gains in ([1.25, 1.5, 1.75, 2.0, 2.25, 2.5, 2.75], [1.0]):
    g = len(gains)
    space = np.stack([np.stack([gn * ui, gn * uq]) for gn in gains])
    ds = ctx.upload(space.reshape(-1))
    bb = [ctx.empty(N // 64 + 2, np.uint64) for _ in range(g)]
    ptrs = (ctypes.c_void_p * g)(*[b.ptr.value for b in bb])
    gs = (ctypes.c_double * g)(*gains)
    args = (ctx.handle, d_f64.ptr, N, xb, dt[0].ptr, dt[1].ptr, dt[2].ptr, dt[3].ptr, ds.ptr, gs, g, mc, dl.ptr, len(lpf), float(np.abs(lpf).sum()), ptrs)
    nlp = 1 if g == 1 else 2
    for label, env in (("fused", None), ("three kernels", "1")):
        ctx.tune(afsk_unfused=1 if env else 0)
        report(f"afsk_sweep_signs_tones g={g} {label} (+ gated fallback launches)",
               timeit(lambda: check(L.pm_afsk_sweep_signs_tones(*args, ctypes.byref(tones)))), (8.0 + g / 8.0) * N, (4 * mc / 16 + 9 + nlp * len(lpf) + g) * N)
        ctx.tune(afsk_unfused=0)
    report(f"afsk_sweep_signs g={g} direct sums (+ gated fallback launches)", timeit(lambda: check(L.pm_afsk_sweep_signs(*args))), (8.0 + g / 8.0) * N,
           (4 * mc + 2 * len(lpf) + g) * N)

# a 10-minute buffer is tens of microseconds at roofline, so the HBM-bound short-tap FIR is also timed on 2^28 samples
# (0.5 GB in, 2.1 GB out per launch) where launch ramp and tail no longer matter
if os.environ.get("KB_BIG", "1") == "1":
    NB = 1 << 28
    big_in = ctx.upload(np.tile(xi[:1 << 20], NB >> 20))
    big_out = ctx.empty(NB, np.float64)
    big_bits = ctx.empty(NB // 64 + 2, np.uint64)
    for m in [8, 40]:
        h = ctx.upload(rng.standard_normal(m))
        report(f"fir_i16 m={m} N=2^28", timeit(lambda: check(L.pm_fir_valid_i16(ctx.handle, big_in.ptr, NB, h.ptr, m, big_out.ptr, 0)), reps=3), 10.0 * NB, m * NB)
    h = ctx.upload(rng.standard_normal(8))
    report("fir_signs_i16 m=8 N=2^28", timeit(lambda: check(L.pm_fir_signs_i16(ctx.handle, big_in.ptr, NB, h.ptr, 8, big_bits.ptr, 0)), reps=3), 2.125 * NB, 8 * NB)
    big_in.free(); big_out.free(); big_bits.free()

if os.environ.get("KB_ONLY") == "fir":
    sys.exit(0)

# slicer on a band-limited stream (realistic crossing density) and on raw noise
sm = np.convolve(rng.standard_normal(N + 39), np.hanning(40), "valid")
for label, arr in [("smooth", sm), ("noise", xi.astype(np.float64))]:
    dx = ctx.upload(arr)
    check(L.pm_signs_f64(ctx.handle, dx.ptr, N, bits.ptr))
    for sps, lock in [(40.0, 0.77), (5.0, 0.88), (160.0, 0.90)]:
        p = SlicerParams()
        p.samples_per_symbol, p.lock_rate, p.bits_per_symbol, p.state_mask = sps, lock, 1, 3
        for k, v in enumerate([0, 0, 1, 1]):
            p.demap[k] = v
        cap = N // 8 + 4
        data, addr = ctx.scratch("kb_d", cap + 4, np.uint8), ctx.scratch("kb_a", cap, np.int64)
        cnt = ctypes.c_int64()
        ms = timeit(lambda: check(L.pm_slice_binary(ctx.handle, bits.ptr, N, ctypes.byref(p), data.ptr, addr.ptr, cap, ctypes.byref(cnt))), reps=3)
        it, cl, nc = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
        L.pm_slicer_stats(ctx.handle, ctypes.byref(it), ctypes.byref(cl), ctypes.byref(nc))
        row = {"kernel": f"slice_binary {label} sps={sps} lock={lock}", "ms_best": round(ms[0], 4), "iterations": it.value,
               "chunk_len": cl.value, "chunks": nc.value, "bytes_out": cnt.value, "Msamples/s": round(N / ms[0] / 1e3, 1)}
        rows.append(row)
        print(json.dumps(row), flush=True)

# sequential recurrences (latency-bound): ns per sample
n2 = min(N, 2_400_000)
buf = ctx.upload(sm[:n2] * 100.0)
st = (ctypes.c_double * 2)(0.0, 0.0)
pa = AGCParams(500.0, 50.0, 1.0, 48000.0, 1.0)
ms = timeit(lambda: check(L.pm_agc_apply(ctx.handle, buf.ptr, n2, ctypes.byref(pa), st)), reps=2)
print(json.dumps({"kernel": "agc", "ms_best": round(ms[0], 3), "ns_per_sample": round(ms[0] * 1e6 / n2, 1)}), flush=True)
import math
tab = ctx.upload(np.array([math.sin(i * 2.0 * math.pi / 256) for i in range(256)]))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymodem_amd import taps as T  # noqa: E402
pd = ctx.upload(np.ascontiguousarray(T.qpsk_error_table().reshape(-1), dtype=np.int32))
x1 = ctx.upload(np.sin(np.arange(n2) * 0.2) * 0.7)
x2 = ctx.upload(np.cos(np.arange(n2) * 0.2) * 0.7)
for nl in [1, 8, 64]:
    loops = (Loop * nl)()
    b0, b1, a1 = T.one_pole_lowpass(48000.0, 250.0, 1.0)
    for k in range(nl):
        lp = loops[k]
        lp.phase_scaling, lp.index_scaling, lp.set_frequency = 2.0 * math.pi / 48000.0, 256 / (2.0 * math.pi), 1500.0 + k
        lp.b0, lp.b1, lp.a1 = b0, b1, a1
        lp.p_rate, lp.i_rate, lp.i_limit, lp.gain = 0.3, 0.3 / 2000, 31.25, 14400 / 65536
    o1 = ctx.scratch("kb_o1", n2 * nl, np.float64)
    o2 = ctx.scratch("kb_o2", n2 * nl, np.float64)
    ms = timeit(lambda: check(L.pm_costas_bpsk(ctx.handle, loops, nl, tab.ptr, x1.ptr, 0, n2, o1.ptr, n2)), reps=2)
    print(json.dumps({"kernel": f"costas x{nl}", "ms_best": round(ms[0], 3), "ns_per_sample": round(ms[0] * 1e6 / n2, 1)}), flush=True)
    ms = timeit(lambda: check(L.pm_mpsk_loop(ctx.handle, loops, nl, tab.ptr, pd.ptr, x1.ptr, x2.ptr, 0, n2, o1.ptr, o2.ptr, n2)), reps=2)
    print(json.dumps({"kernel": f"mpsk_loop x{nl}", "ms_best": round(ms[0], 3), "ns_per_sample": round(ms[0] * 1e6 / n2, 1)}), flush=True)
