"""GPU parity: every HIP kernel, through the C ABI, against the oracle on the same seeded inputs.
Bar: bit-exact (the oracle's *_canon FIRs use the kernels' summation order; the loops and slicers are
sequential restatements), so np.array_equal on float64 arrays, bytes and addresses."""
import ctypes

import numpy as np
import pytest

from conftest import noise_i16, tuned
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import pymodem_amd
    if pymodem_amd.lib().pm_device_count() < 1:
        pytest.fail("no GPU visible: -m gpu tests need an MI355X")
    return pymodem_amd.Context.default()


def L():
    import pymodem_amd
    return pymodem_amd.lib()


def chk(rc):
    from pymodem_amd._native import check
    check(rc)


def fir_gpu(ctx, x, h, flags=0):
    dx = ctx.upload(x)
    dh = ctx.upload(np.ascontiguousarray(h, dtype=np.float64))
    y = ctx.empty(len(x) - len(h) + 1, np.float64)
    fn = L().pm_fir_valid_i16 if x.dtype == np.int16 else L().pm_fir_valid_f64
    chk(fn(ctx.handle, dx.ptr, len(x), dh.ptr, len(h), y.ptr, flags))
    return y.download()


@pytest.mark.parametrize("m", [1, 2, 7, 8, 9, 40, 60, 100, 148, 241, 961, 1120])
def test_fir_bit_exact(ctx, m):
    rng = np.random.default_rng(m)
    h = rng.standard_normal(m)
    for n in sorted({m, m + 1, m + 2047, m + 2048, m + 2049, 5000 + m, 70001}):
        if n < m:
            continue
        xi = np.clip(np.rint(rng.standard_normal(n) * 8000), -32768, 32767).astype(np.int16)
        assert np.array_equal(fir_gpu(ctx, xi, h), O.fir_canon(xi, h)), (m, n, "i16")
        xf = rng.standard_normal(n) * 100.0
        assert np.array_equal(fir_gpu(ctx, xf, h), O.fir_canon(xf, h)), (m, n, "f64")
    xf = rng.standard_normal(4096 + m)
    assert np.array_equal(fir_gpu(ctx, xf, h, flags=1), -O.fir_canon(xf, h))
    # and the reference's own summation order agrees to rounding
    ref = np.convolve(xf, h, "valid")
    assert np.abs(fir_gpu(ctx, xf, h) - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-300)


def fir_limbs_gpu(ctx, x, h):
    dx = ctx.upload(x)
    hh = np.ascontiguousarray(h, dtype=np.float64)
    y = ctx.empty(len(x) - len(h) + 1, np.float64)
    bound = ctypes.c_double(-1.0)
    chk(L().pm_fir_valid_i16_limbs(ctx.handle, dx.ptr, len(x), hh.ctypes.data, len(h), y.ptr, ctypes.byref(bound)))
    return y.download(), bound.value


@pytest.mark.parametrize("m", [1, 16, 50, 148, 177, 178, 241])
def test_fir_int8_limbs_within_its_bound(ctx, m):
    """The band-pass on the int8 matrix pipe (pm_bpf8.hip): exact integer products of quantised taps, so the distance from the
    reference's sum is the quantisation + a few roundings, and the entry point states it.  Checked against an integer-exact sum
    (Python ints / float128-free: the taps as exact fractions), against the bit-exact kernel, and on the inputs that stress the
    digit split: both ends of the int16 range, alternating signs, ragged lengths around the 4096-output workgroups."""
    from fractions import Fraction
    rng = np.random.default_rng(900 + m)
    from pymodem_amd import taps as T
    h = T.windowed_sinc(m, [900.0, 2500.0], 48000.0, False) if m >= 50 else rng.standard_normal(m) * 0.1      # afsk.py:126-132 / random
    h = np.asarray(h, np.float64)
    scale = float(np.abs(h).sum()) * 32768.0
    cases = []
    for n in sorted({m, m + 1, m + 255, m + 4095, m + 4096, m + 4097, 3 * 4096 + m + 17, 70001}):
        cases.append(np.clip(np.rint(rng.standard_normal(n) * 8000), -32768, 32767).astype(np.int16))
    n = 9000 + m
    cases += [np.full(n, 32767, np.int16), np.full(n, -32768, np.int16), np.zeros(n, np.int16),
              np.where(np.arange(n) % 2 == 0, 32767, -32768).astype(np.int16), rng.integers(-32768, 32768, n).astype(np.int16),
              np.where(np.sign(h[::-1])[np.arange(n) % m] >= 0, 32767, -32768).astype(np.int16)]      # lines up with the taps: the largest sums
    for x in cases:
        y, bound = fir_limbs_gpu(ctx, x, h)
        assert 0.0 < bound <= 1e-8 * scale, (m, bound, scale)      # (32-bit taps: ~2^-31 of the largest sum)
        ref = O.fir_canon(x, h)
        assert np.array_equal(fir_gpu(ctx, x, h), ref)
        assert np.abs(y - ref).max() <= bound, (m, len(x), np.abs(y - ref).max(), bound)
    # a handful of outputs against exact rational arithmetic: the bound holds against the true sum too (minus the reference's own share)
    x = cases[-1]
    y, bound = fir_limbs_gpu(ctx, x, h)
    hf = [Fraction(float(v)) for v in h]
    for k in (0, 1, len(y) // 2, len(y) - 1):
        exact = sum(hf[m - 1 - t] * int(x[k + t]) for t in range(m))
        assert abs(Fraction(float(y[k])) - exact) <= Fraction(bound), (m, k)


def test_fir_int8_limbs_rejects_what_it_cannot_hold(ctx):
    x = ctx.upload(noise_i16(5000, 3))
    y = ctx.empty(5000, np.float64)
    h = np.zeros(242)
    h[0] = 1.0
    b = ctypes.c_double()
    assert L().pm_fir_valid_i16_limbs(ctx.handle, x.ptr, 5000, h.ctypes.data, 242, y.ptr, ctypes.byref(b)) != 0      # more than 241 taps
    z = np.zeros(8)
    assert L().pm_fir_valid_i16_limbs(ctx.handle, x.ptr, 5000, z.ctypes.data, 8, y.ptr, ctypes.byref(b)) != 0        # all-zero taps
    assert L().pm_fir_valid_i16_limbs(ctx.handle, x.ptr, 5, h.ctypes.data, 8, y.ptr, ctypes.byref(b)) != 0            # fewer samples than taps
    assert L().pm_fir_valid_i16_limbs(ctx.handle, ctypes.c_void_p(x.ptr.value + 2), 4000, h.ctypes.data, 8, y.ptr, ctypes.byref(b)) != 0     # input not 16-byte aligned


@pytest.mark.parametrize("exponent", [-60, -21, -1, 0, 1, 24, 59, 100])
def test_v_sqrt_f32_is_within_one_ulp(ctx, exponent):
    """slide_run_f32 (csrc/pm_fir.hip) takes the roots of the certified sweeps' magnitudes with v_sqrt_f32 and prices the instruction at
    one unit in the last place in its bound: here the device evaluates it on every one of the 2^24 binary32 values of two neighbouring
    binades (a root's significand depends on the radicand's significand and its exponent's parity only) against the correctly rounded
    binary64 root."""
    worst = ctypes.c_int64(-1)
    chk(L().pm_ubench_sqrt_f32(ctx.handle, exponent, ctypes.byref(worst)))
    assert 0 < worst.value <= 1024, worst.value / 1024.0


def rows_max_gpu(ctx, rows2d, h):
    rows, n = rows2d.shape
    stride = (n + 7) // 8 * 8
    flat = np.zeros((rows, stride), np.int16)
    flat[:, :n] = rows2d
    dx = ctx.upload(flat.reshape(-1))
    hh = np.ascontiguousarray(h, dtype=np.float64)
    out = np.full(rows, np.nan)
    redone = ctypes.c_int64(-1)
    chk(L().pm_bpf8_rows_max_i16(ctx.handle, dx.ptr, stride, rows, n, hh.ctypes.data, len(h), out.ctypes.data, ctypes.byref(redone)))
    return out, redone.value


@pytest.mark.parametrize("m", [1, 16, 148, 177, 178, 240, 241])
def test_band_pass_maximum_is_the_reference_sums_maximum(ctx, m):
    """AGC.apply's `normal` = max(band-passed recording) (agc.py:67, psk.py:165-168) without the band-passed recording
    (pm_bpf8.hip: bpf8_max_kernel): values from the matrix pipe choose the candidates, the canonical chain decides -- the result is
    max() of the oracle's FIR bit for bit, on noise, tones, silence (whole and partial), full-scale constants, the alignment
    that makes the largest sums, periodic inputs whose maximum repeats thousands of times, and ragged lengths."""
    from pymodem_amd import taps as T
    rng = np.random.default_rng(40 + m)
    h = np.asarray(T.windowed_sinc(m, [1200.0, 1800.0], 48000.0, False) if m >= 50 else rng.standard_normal(m) * 0.1, np.float64)
    for n in (m, m + 1, m + 4095, m + 4096, m + 4097, 5 * 4096 + m + 3, 150001):
        t = np.arange(n)
        rows = [
            np.clip(np.rint(rng.standard_normal(n) * 6000), -32768, 32767),
            np.rint(9000 * np.sin(2 * np.pi * 1500.0 * t / 48000.0) + 200 * rng.standard_normal(n)),
            np.zeros(n),
            np.where((t // 5000) % 2 == 0, 0, np.rint(rng.standard_normal(n) * 3000)),
            np.full(n, 32767), np.full(n, -32768),
            np.where(np.sign(h[::-1])[t % m] >= 0, 32767, -32768),
            np.rint(12000 * np.sin(2 * np.pi * (t % 32) / 32.0)),                      # 1500 Hz exactly: every period the same sums
            np.where(t < n // 2, np.rint(rng.standard_normal(n) * 100), 0),
            -np.abs(np.rint(rng.standard_normal(n) * 50)) * (np.abs(h).sum() > 0),
        ]
        x = np.stack(rows).astype(np.int16)
        got, redone = rows_max_gpu(ctx, x, h)
        want = np.array([O.fir_canon(r, h).max() for r in x])
        assert got.tobytes() == want.tobytes(), (m, n, got, want)
        assert redone >= 1
    # what goes through the exact chain on a recording-like input is a handful of outputs per row
    n = 2_000_000
    x = np.clip(np.rint(rng.standard_normal((4, n)) * 4000), -32768, 32767).astype(np.int16)
    got, redone = rows_max_gpu(ctx, x, h)
    assert got.tobytes() == np.array([O.fir_canon(r, h).max() for r in x]).tobytes()
    assert redone < 4 * 4000, redone                            # (at most about one per workgroup of the first wave of workgroups)


def test_fir_rejects_bad_arguments(ctx):
    from pymodem_amd import NativeError
    x = ctx.upload(np.zeros(4))
    h = ctx.upload(np.ones(8))
    y = ctx.empty(8, np.float64)
    with pytest.raises(NativeError):
        chk(L().pm_fir_valid_f64(ctx.handle, x.ptr, 4, h.ptr, 8, y.ptr, 0))     # n < m
    with pytest.raises(NativeError):
        chk(L().pm_fir_valid_f64(ctx.handle, None, 16, h.ptr, 8, y.ptr, 0))


@pytest.mark.parametrize("m", [3, 4, 8, 14, 40, 60, 61])
def test_afsk_correlate_bit_exact(ctx, m):
    rng = np.random.default_rng(100 + m)
    taps = [rng.standard_normal(m) for _ in range(4)]
    for n in [m, m + 1023, m + 1024, m + 1025, 50000]:
        x = rng.standard_normal(n) * 300.0
        dx = ctx.upload(x)
        dt = [ctx.upload(t) for t in taps]
        y = ctx.empty(n - m + 1, np.float64)
        chk(L().pm_afsk_correlate(ctx.handle, dx.ptr, n, dt[0].ptr, dt[1].ptr, dt[2].ptr, dt[3].ptr, m, y.ptr))
        assert np.array_equal(y.download(), O.afsk_correlate_canon(x, *taps)), (m, n)


@pytest.mark.parametrize("groups", [1, 2, 3, 7, 8])
@pytest.mark.parametrize("m", [1, 2, 3, 5, 40, 60, 61])
def test_afsk_correlate_group_bit_exact(ctx, groups, m):
    """Shared-mark correlator banks: every output stream equals the single-modem kernel's and the oracle's."""
    rng = np.random.default_rng(1000 * groups + m)
    mark = [rng.standard_normal(m) for _ in range(2)]
    space = rng.standard_normal((groups, 2, m))
    for n, aligned in [(m, True), (m + 511, True), (m + 512, False), (m + 513, True), (40000, True), (40001, False)]:
        x = rng.standard_normal(n) * 300.0
        dx = ctx.upload(x)
        dm = [ctx.upload(t) for t in mark]
        ds = ctx.upload(space.reshape(-1))
        nout = n - m + 1
        stride = (nout + 63) // 64 * 64 if aligned else nout + 1          # odd stride: the scalar-store path
        y = ctx.empty(stride * groups, np.float64)
        chk(L().pm_afsk_correlate_group(ctx.handle, dx.ptr, n, dm[0].ptr, dm[1].ptr, ds.ptr, groups, m, y.ptr, stride))
        got = y.download()
        for g in range(groups):
            want = O.afsk_correlate_canon(x, mark[0], mark[1], space[g, 0], space[g, 1])
            assert np.array_equal(got[g * stride:g * stride + nout], want), (groups, m, n, g)


def test_afsk_correlate_group_rejects_bad_arguments(ctx):
    from pymodem_amd import NativeError
    d = ctx.upload(np.zeros(64))
    with pytest.raises(NativeError):
        chk(L().pm_afsk_correlate_group(ctx.handle, d.ptr, 64, d.ptr, d.ptr, d.ptr, 9, 4, d.ptr, 64))
    with pytest.raises(NativeError):
        chk(L().pm_afsk_correlate_group(ctx.handle, d.ptr, 64, d.ptr, d.ptr, d.ptr, 2, 4, d.ptr, 10))      # streams would overlap


def test_sqrt_is_correctly_rounded(ctx):
    """The correlator magnitude relies on the device sqrt being IEEE: one tap turns the kernel into sqrt(a*a + 0)."""
    rng = np.random.default_rng(5)
    x = np.abs(rng.standard_normal(200000)) * 10.0 ** rng.integers(-150, 150, 200000)
    one, zero = ctx.upload(np.ones(1)), ctx.upload(np.zeros(1))
    dx = ctx.upload(x)
    y = ctx.empty(len(x), np.float64)
    chk(L().pm_afsk_correlate(ctx.handle, dx.ptr, len(x), one.ptr, zero.ptr, zero.ptr, zero.ptr, 1, y.ptr))
    assert np.array_equal(y.download(), np.sqrt(x * x + 0.0 * 0.0) - np.sqrt(0.0))


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 65536, 100003])
def test_signs(ctx, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n)
    x[::7] = 0.0
    x[3::11] = -0.0
    dx = ctx.upload(x)
    bits = ctx.empty((n + 63) // 64, np.uint64)
    chk(L().pm_signs_f64(ctx.handle, dx.ptr, n, bits.ptr))
    got = np.unpackbits(bits.download().view(np.uint8), bitorder="little")[:n].astype(bool)
    assert np.array_equal(got, x >= 0)


def slicer_input(n, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.standard_normal(n)
    if kind == "smooth":          # band-limited: few crossings per symbol, like a real demodulated stream
        return np.convolve(rng.standard_normal(n + 63), np.hanning(64), "valid")
    if kind == "silence":         # exact zeros: no crossing ever, the fixed point needs one iteration per chunk
        return np.zeros(n)
    if kind == "alternating":     # a crossing at every sample
        return np.where(np.arange(n) % 2 == 0, 1.0, -1.0)
    raise ValueError(kind)


BIN = [(48000, "1200", "0.77"), (8000, "300", "0.90"), (44100, "1200", "0.75"), (48000, "9600", "0.88"), (48000, "300", "0.99")]


@pytest.mark.parametrize("kind", ["noise", "smooth", "silence", "alternating"])
@pytest.mark.parametrize("rate,cfg,lock", BIN)
def test_binary_slicer_bit_exact(ctx, rate, cfg, lock, kind):
    from pymodem_amd.slicer import BinarySlicer
    for n in [1, 100, 1023, 1024, 1025, 40000, 300001]:
        if kind == "silence" and n > 40000:
            continue
        x = slicer_input(n, n + rate, kind)
        s = BinarySlicer(sample_rate=rate, config=cfg)
        s.StringOptionsRetune({"lock_rate": lock})
        got = s.slice(x)
        d, a = O.BinarySlicer(rate, cfg, {"lock_rate": lock}).slice(x)
        assert np.array_equal(got.data, d) and np.array_equal(got.address, a), (n, kind, s.last_stats)


QUAD = [(48000, "qpsk_2400", "0.98"), (8000, "qpsk_600", "0.815"), (48000, "bpsk_1200", "0.9"), (44100, "qpsk_3600", "0.985"),
        (48000, "bpsk_300", "0.815")]


@pytest.mark.parametrize("kind", ["noise", "smooth", "alternating"])
@pytest.mark.parametrize("rate,cfg,lock", QUAD)
def test_quadrature_slicer_bit_exact(ctx, rate, cfg, lock, kind):
    from pymodem_amd.data_classes import IQData
    from pymodem_amd.slicer import QuadratureSlicer
    for n in [1, 777, 4096, 50001, 300000]:
        iq = IQData()
        iq.i_data = slicer_input(n, n + rate, kind)
        iq.q_data = slicer_input(n, n + rate + 1, "smooth" if kind == "alternating" else kind)
        s = QuadratureSlicer(sample_rate=rate, config=cfg)
        s.StringOptionsRetune({"lock_rate": lock})
        got = s.slice(iq)
        d, a = O.QuadratureSlicer(rate, cfg, {"lock_rate": lock}).slice((iq.i_data, iq.q_data))
        assert np.array_equal(got.data, d) and np.array_equal(got.address, a), (n, kind, s.last_stats)


@pytest.mark.parametrize("kind", ["noise", "smooth", "alternating"])
def test_slicers_continue_across_calls(ctx, kind):
    """A slicer object carries phase clock, last sample sign, the open byte and the address count from one slice() to the next
    (slicer.py:49-56): cutting a stream anywhere and slicing the pieces one after another gives the uncut result."""
    from pymodem_amd.data_classes import IQData
    from pymodem_amd.slicer import BinarySlicer, QuadratureSlicer
    rng = np.random.default_rng(42)
    n = 120000
    xi, xq = slicer_input(n, 11, kind), slicer_input(n, 12, "smooth" if kind == "alternating" else kind)
    for trial in range(4):
        cuts = np.unique(np.concatenate([[0, n], rng.integers(1, n, 6), [1, 63, 64, 65, 4097][:trial + 1]]))
        # binary
        s = BinarySlicer(sample_rate=48000, config="1200")
        s.StringOptionsRetune({"lock_rate": "0.77"})
        o = O.BinarySlicer(48000, "1200", {"lock_rate": "0.77"})
        whole_d, whole_a = O.BinarySlicer(48000, "1200", {"lock_rate": "0.77"}).slice(xi)
        parts_d, parts_a = [], []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            got = s.slice(xi[lo:hi])
            od, oa = o.slice(xi[lo:hi])
            assert np.array_equal(got.data, od) and np.array_equal(got.address, oa), (kind, lo, hi)
            assert s._state.phase_clock == o.state[0] and s._state.working_bits == int(o.state[4]) and s._state.streamaddress == int(o.state[5])
            parts_d.append(got.data)
            parts_a.append(got.address)
        assert np.array_equal(np.concatenate(parts_d), whole_d) and np.array_equal(np.concatenate(parts_a), whole_a)
        # quadrature, differential demap across the cut
        q = QuadratureSlicer(sample_rate=48000, config="qpsk_2400")
        oq = O.QuadratureSlicer(48000, "qpsk_2400", {})
        whole_d, whole_a = O.QuadratureSlicer(48000, "qpsk_2400", {}).slice((xi, xq))
        parts_d, parts_a = [], []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            iq = IQData()
            iq.i_data, iq.q_data = xi[lo:hi], xq[lo:hi]
            got = q.slice(iq)
            od, oa = oq.slice((xi[lo:hi], xq[lo:hi]))
            assert np.array_equal(got.data, od) and np.array_equal(got.address, oa), (kind, lo, hi)
            parts_d.append(got.data)
            parts_a.append(got.address)
        assert np.array_equal(np.concatenate(parts_d), whole_d) and np.array_equal(np.concatenate(parts_a), whole_a)
    # output buffers are sized for 2.2x the nominal symbol count; when a stream exceeds that the batch is redone with the hard bound
    import pymodem_amd.slicer as slicer_mod
    s.tune()
    slicer_mod._TIGHT_FACTOR = 0.001
    try:
        a1, a2 = s.slice(xi[:70000]), s.slice(xi[70000:])
    finally:
        slicer_mod._TIGHT_FACTOR = 2.2
    whole = O.BinarySlicer(48000, "1200", {"lock_rate": "0.77"}).slice(xi)
    assert np.array_equal(np.concatenate([a1.data, a2.data]), whole[0]) and np.array_equal(np.concatenate([a1.address, a2.address]), whole[1])
    s.tune()                                   # retuning returns the object to the just-built state
    again = s.slice(xi)
    fresh = O.BinarySlicer(48000, "1200", {"lock_rate": "0.77"}).slice(xi)
    assert np.array_equal(again.data, fresh[0]) and np.array_equal(again.address, fresh[1])


def test_agc_bit_exact(ctx, golden):
    from pymodem_amd._native import AGCParams
    g = golden("primitives")
    cases = [("agc_8k_in", 8000.0, 0.5), ("agc_48k_in", 48000.0, 0.5), ("agc_neg_in", 8000.0, 0.01), ("agc_zero_in", 8000.0, 0.01)]
    for key, rate, sustain in cases:
        x = g[key].copy()
        d = ctx.upload(x)
        st = (ctypes.c_double * 2)(0.0, 0.0)
        p = AGCParams(500.0, 50.0, sustain, rate, 1.0)
        chk(L().pm_agc_apply(ctx.handle, d.ptr, len(x), ctypes.byref(p), st))
        want = x.copy()
        ost = np.zeros(2)
        O.agc_apply(want, rate, 500.0, sustain, 50.0, 1.0, state=ost)
        assert np.array_equal(d.download(), want), key
        assert np.array_equal(d.download(), g[key.replace("_in", "_out")]), key      # and the reference itself
        assert st[0] == ost[0] and st[1] == ost[1]


def loops_pair(rate, carrier, cutoff, p, i, lim, gain, integral0=0.0):
    from pymodem_amd._native import Loop
    a = O.make_loop(rate, carrier, cutoff, 1.0, p, i, lim, gain, integral0)
    b = Loop()
    for f, _ in a._fields_:                # the product's pm_loop has these and the QPSK branch-filter fields after them
        setattr(b, f, getattr(a, f))
    return a, b


def agc_like(n, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    return np.sin(2 * np.pi * 1503.7 * t / 48000 + 0.4) * (0.6 + 0.3 * np.sin(t / 5000.0)) + 0.2 * rng.standard_normal(n)


@pytest.mark.parametrize("n", [1, 255, 256, 257, 20000])
def test_costas_and_pll_bit_exact(ctx, n):
    tab = O.nco_table()
    dt = ctx.upload(tab)
    x = agc_like(n, n)
    dx = ctx.upload(x)
    for fn, ofn, args in [(L().pm_costas_bpsk, O.costas_bpsk, (48000.0, 1500.0, 250.0, 0.06, 0.06 / 1000, 31.25, 7200)),
                          (L().pm_pll_afsk, O.pll_afsk, (8000.0, 1700.0, 150.0, 0.6, 0.6 / 6000, 50, 900))]:
        a, b = loops_pair(*args)
        out = ctx.empty(n, np.float64)
        chk(fn(ctx.handle, ctypes.byref(b), 1, dt.ptr, dx.ptr, 0, n, out.ptr, n))
        want = ofn(a, x, tab)
        assert np.array_equal(out.download(), want)
        for f, _ in a._fields_:
            assert getattr(a, f) == getattr(b, f), f           # end state identical too


@pytest.mark.parametrize("n", [1, 255, 256, 257, 20000])
def test_costas_qpsk_bit_exact(ctx, n):
    """QPSKModem's loop (psk.py:434-467): both arms, loop state and branch-filter state against the oracle; 5 loops in one launch."""
    tab = O.nco_table()
    dt = ctx.upload(tab)
    x = agc_like(n, 3 * n + 1)
    dx = ctx.upload(x)
    from pymodem_amd._native import Loop
    nl = 5
    loops = (Loop * nl)()
    want = []
    for k in range(nl):
        a, b = loops_pair(48000.0, 1800.0 + 3.0 * k, 200.0, 0.1, 0.1 / 500, 87.5, 450.0)
        br = np.array(list(O.iir1_coefs(48000.0, 1200.0, 1.0)) + [0.0] * 6)
        b.bb0, b.bb1, b.ba1 = br[:3]
        ctypes.memmove(ctypes.byref(loops[k]), ctypes.byref(b), ctypes.sizeof(Loop))
        oi, oq = O.costas_qpsk(a, br, x, tab)
        want.append((oi, oq, a, br))
    oi_d, oq_d = ctx.empty(n * nl, np.float64), ctx.empty(n * nl, np.float64)
    chk(L().pm_costas_qpsk(ctx.handle, loops, nl, dt.ptr, dx.ptr, 0, n, oi_d.ptr, oq_d.ptr, n))
    gi, gq = oi_d.download().reshape(nl, n), oq_d.download().reshape(nl, n)
    for k, (oi, oq, a, br) in enumerate(want):
        assert np.array_equal(gi[k], oi) and np.array_equal(gq[k], oq), k
        for f, _ in a._fields_:
            assert getattr(a, f) == getattr(loops[k], f), f
        assert [loops[k].cx0, loops[k].cx1, loops[k].cy0, loops[k].sx0, loops[k].sx1, loops[k].sy0] == list(br[3:])


def test_mpsk_loop_batch_bit_exact(ctx):
    """Eleven loops at swept carriers over one shared input (the qpsk_2400.json arrangement), one launch."""
    from pymodem_amd._native import Loop
    n = 30000
    tab, pdt = O.nco_table(), O.pd_table()
    re, im = agc_like(n, 1), agc_like(n, 2)
    carriers = [1450.0 + 10 * k for k in range(11)]
    pairs = [loops_pair(48000.0, c, 250.0, 0.3, 0.3 / 2000, 31.25, 14400 / 65536, -31.25) for c in carriers]
    arr = (Loop * len(pairs))(*[b for _, b in pairs])
    io, qo = ctx.empty(n * len(pairs), np.float64), ctx.empty(n * len(pairs), np.float64)
    dtab, dpd, dre, dim = ctx.upload(tab), ctx.upload(np.ascontiguousarray(pdt.reshape(-1), dtype=np.int32)), ctx.upload(re), ctx.upload(im)
    chk(L().pm_mpsk_loop(ctx.handle, arr, len(pairs), dtab.ptr, dpd.ptr, dre.ptr, dim.ptr, 0, n, io.ptr, qo.ptr, n))
    gi, gq = io.download().reshape(len(pairs), n), qo.download().reshape(len(pairs), n)
    for k, (a, _) in enumerate(pairs):
        wi, wq = O.mpsk_loop(a, re, im, tab, pdt)
        assert np.array_equal(gi[k], wi) and np.array_equal(gq[k], wq), k
        assert a.phase == arr[k].phase and a.integral == arr[k].integral and a.control == arr[k].control


@pytest.mark.parametrize("m", [1, 8, 13, 100, 241])
def test_fir_signs_fused(ctx, m):
    """pm_fir_signs_*: the bitmap equals (FIR output >= 0) of the unfused kernel / the oracle, for both input types and NEGATE."""
    rng = np.random.default_rng(300 + m)
    h = rng.standard_normal(m)
    dh = ctx.upload(h)
    for n in [m, m + 63, m + 64, m + 2047, m + 2048, 70003]:
        xi = np.clip(np.rint(rng.standard_normal(n) * 8000), -32768, 32767).astype(np.int16)
        xf = rng.standard_normal(n)
        xf[::5] = 0.0                                   # exact zeros in the output are ">= 0"
        for x, fn in [(xi, L().pm_fir_signs_i16), (xf, L().pm_fir_signs_f64)]:
            for flags in (0, 1):
                nout = n - m + 1
                bits = ctx.empty((nout + 63) // 64 + 1, np.uint64)
                dx = ctx.upload(x)
                chk(fn(ctx.handle, dx.ptr, n, dh.ptr, m, bits.ptr, flags))
                got = np.unpackbits(bits.download().view(np.uint8), bitorder="little")[:nout].astype(bool)
                y = O.fir_canon(x, h)
                want = (-y if flags else y) >= 0
                assert np.array_equal(got, want), (m, n, x.dtype, flags)


def _signs(ctx, fn, x, h, flags=0):
    n, m = len(x), len(h)
    nout = n - m + 1
    bits = ctx.empty((nout + 63) // 64 + 1, np.uint64)
    dx, dh = ctx.upload(x), ctx.upload(h)
    chk(fn(ctx.handle, dx.ptr, n, dh.ptr, m, bits.ptr, flags))
    words = bits.download()
    got = np.unpackbits(words.view(np.uint8), bitorder="little")
    assert not got[nout:((nout + 63) // 64) * 64].any()                  # padding bits of the last word are zero
    return got[:nout].astype(bool)


@pytest.mark.parametrize("m", [16, 17, 40, 100, 241, 961])
def test_fir_signs_on_the_decision_boundary(ctx, m):
    """The fused sign output on inputs made to sit ON the decision boundary: outputs that are exactly zero, tiny against their
    neighbours, sums that cancel, huge dynamic range, denormals, values beyond binary32, non-finite samples -- every bit must be the
    sign of the canonical binary64 sum (any reduced-precision shortcut would show here)."""
    rng = np.random.default_rng(900 + m)
    n = 3 * 4096 + m + 77
    h_rand = rng.standard_normal(m)
    h_anti = h_rand - h_rand[::-1]                                       # antisymmetric: symmetric inputs give exact zeros
    cases = []
    base = rng.standard_normal(n)
    cases.append(("noise", base, h_rand))
    sym = np.concatenate([base[:n // 2], base[:n - n // 2][::-1]])
    cases.append(("cancelling", np.round(sym * 64) / 64, np.round(h_anti * 64) / 64))      # exact products: true zeros appear
    quiet = base * 1e-9
    quiet[::4096] = 1e6                                                  # one huge sample per tile: E is large, most outputs fall back
    cases.append(("dynamic range", quiet, h_rand))
    cases.append(("zeros", np.zeros(n), h_rand))
    cases.append(("denormal", base * 1e-310, h_rand))
    cases.append(("beyond binary32", base * 1e200, h_rand))
    cases.append(("tiny taps", base, h_rand * 1e-60))
    nf = base.copy()
    nf[1000], nf[5000], nf[9000] = np.inf, -np.inf, np.nan
    cases.append(("non-finite", nf, h_rand))
    slow = np.sin(np.arange(n) * 0.002) * 1e4 + 1e-3 * rng.standard_normal(n)       # many outputs close to zero at each crossing
    cases.append(("slow sine", slow, np.ones(m) / m))
    for name, x, h in cases:
        for flags in (0, 1):
            got = _signs(ctx, L().pm_fir_signs_f64, x, h, flags)
            with np.errstate(all="ignore"):
                y = O.fir_canon(x, h)
                want = (-y if flags else y) >= 0
            assert np.array_equal(got, want), (name, m, flags, int(np.count_nonzero(got != want)))
    xi = np.clip(np.rint(base * 8000), -32768, 32767).astype(np.int16)
    xi[2000:2600] = 0
    for flags in (0, 1):
        got = _signs(ctx, L().pm_fir_signs_i16, xi, h_anti, flags)
        y = O.fir_canon(xi, h_anti)
        assert np.array_equal(got, (-y if flags else y) >= 0), ("int16", m, flags)


def test_fir_signs_full_size_matches_exact_kernel(ctx):
    """28.8 M samples of an AFSK-correlator-like stream: the fused bitmap against (unfused FIR output >= 0)."""
    rng = np.random.default_rng(4)
    n = 28_800_000
    t = np.arange(n)
    x = np.sin(2 * np.pi * t / 40.0 + 3.0 * np.sin(t / 9000.0)) * (1.0 + 0.5 * np.sin(t / 70000.0)) * 3e5 + 2e4 * rng.standard_normal(n)
    from pymodem_amd import taps as T
    h = T.windowed_sinc(100, 900.0, 48000.0, pass_zero=True)
    dx, dh = ctx.upload(x), ctx.upload(h)
    nout = n - len(h) + 1
    y = ctx.empty(nout, np.float64)
    chk(L().pm_fir_valid_f64(ctx.handle, dx.ptr, n, dh.ptr, len(h), y.ptr, 0))
    bits = ctx.empty((nout + 63) // 64 + 1, np.uint64)
    chk(L().pm_fir_signs_f64(ctx.handle, dx.ptr, n, dh.ptr, len(h), bits.ptr, 0))
    got = np.unpackbits(bits.download().view(np.uint8), bitorder="little")[:nout].astype(bool)
    assert np.array_equal(got, y.download() >= 0)


def test_fir_limits_and_unaligned_buffers(ctx):
    """Largest supported filter (8192 taps: the LDS image needs the raised dynamic-LDS limit), one tap more is refused, and device
    pointers that are not 16-byte aligned take the scalar load/store paths of every FIR-class kernel with the same bits."""
    from pymodem_amd import NativeError
    rng = np.random.default_rng(8192)
    h = rng.standard_normal(8192)
    x = rng.standard_normal(8192 + 3000)
    assert np.array_equal(fir_gpu(ctx, x, h), O.fir_canon(x, h))
    xi = np.clip(np.rint(rng.standard_normal(8192 + 700) * 8000), -32768, 32767).astype(np.int16)
    assert np.array_equal(fir_gpu(ctx, xi, h), O.fir_canon(xi, h))
    dx, dh, dy = ctx.upload(x), ctx.upload(np.ones(8193)), ctx.empty(len(x), np.float64)
    with pytest.raises(NativeError):
        chk(L().pm_fir_valid_f64(ctx.handle, dx.ptr, len(x), dh.ptr, 8193, dy.ptr, 0))
    # unaligned: inputs and outputs start one element into their allocations
    m, n = 45, 9000
    h = rng.standard_normal(m)
    dh = ctx.upload(h)
    xf = rng.standard_normal(n + 1)
    big_in, big_out = ctx.upload(xf), ctx.empty(n + 2, np.float64)
    vin, vout = big_in.view(1, n), big_out.view(1, n - m + 1)
    chk(L().pm_fir_valid_f64(ctx.handle, vin.ptr, n, dh.ptr, m, vout.ptr, 0))
    assert np.array_equal(vout.download(), O.fir_canon(xf[1:], h))
    xs = np.clip(np.rint(rng.standard_normal(n + 1) * 8000), -32768, 32767).astype(np.int16)
    big_i = ctx.upload(xs)
    chk(L().pm_fir_valid_i16(ctx.handle, big_i.view(1, n).ptr, n, dh.ptr, m, vout.ptr, 0))
    assert np.array_equal(vout.download(), O.fir_canon(xs[1:], h))
    t = [ctx.upload(rng.standard_normal(m)) for _ in range(4)]
    chk(L().pm_afsk_correlate(ctx.handle, vin.ptr, n, t[0].ptr, t[1].ptr, t[2].ptr, t[3].ptr, m, vout.ptr))
    assert np.array_equal(vout.download(), O.afsk_correlate_canon(xf[1:], *[b.download() for b in t]))
    bits = ctx.empty((n - m + 1 + 63) // 64 + 1, np.uint64)
    chk(L().pm_fir_signs_f64(ctx.handle, vin.ptr, n, dh.ptr, m, bits.ptr, 0))
    got = np.unpackbits(bits.download().view(np.uint8), bitorder="little")[:n - m + 1].astype(bool)
    assert np.array_equal(got, O.fir_canon(xf[1:], h) >= 0)


def test_slice_batch_many_streams_and_ragged_lengths(ctx):
    """More streams than one pm_slice_batch call takes (64): the host splits the batch; lengths from 1 sample to a few thousand,
    binary and quadrature mixed, an empty stream in the middle -- each equals its own oracle run."""
    from pymodem_amd.data_classes import IQData
    from pymodem_amd.slicer import BinarySlicer, QuadratureSlicer, slice_batch
    rng = np.random.default_rng(2024)
    slicers, inputs, wants = [], [], []
    for k in range(70):
        n = int(rng.integers(1, 6000)) if k != 33 else 1
        xi, xq = slicer_input(n, 500 + k, "smooth"), slicer_input(n, 900 + k, "smooth")
        if k % 3 == 0:
            s = QuadratureSlicer(sample_rate=48000, config="qpsk_2400")
            iq = IQData()
            iq.i_data, iq.q_data = xi, xq
            inputs.append(iq)
            wants.append(O.QuadratureSlicer(48000, "qpsk_2400", {}).slice((xi, xq)))
        else:
            s = BinarySlicer(sample_rate=48000, config="9600" if k % 2 else "1200")
            inputs.append(xi)
            wants.append(O.BinarySlicer(48000, "9600" if k % 2 else "1200", {}).slice(xi))
        slicers.append(s)
    bitmaps = [s.sign_bitmaps(x) for s, x in zip(slicers, inputs)]
    got = slice_batch(slicers, bitmaps)
    for k, (g, w) in enumerate(zip(got, wants)):
        assert np.array_equal(g.data, w[0]) and np.array_equal(g.address, w[1]), k


def test_slice_batch_compact_form_equals_the_plain_one(ctx):
    """defer + compact (pm_slice_compact: exact counts, 16-bit address steps; what the pipelined executor brings to the host) gives
    the arrays of the plain call -- ragged lengths, an empty stream, quadrature and binary mixed, and a symbol rate so low that
    eight symbol periods do not fit 16 bits (that stream's addresses then come in full)."""
    from pymodem_amd.data_classes import IQData
    from pymodem_amd.slicer import BinarySlicer, QuadratureSlicer, slice_batch
    from pymodem_amd import chain_execute as ce
    from pymodem_amd import lfsr as L, chain_builder as cb

    def build():
        rng = np.random.default_rng(77)
        slicers, inputs = [], []
        for k in range(12):
            n = [1, 7, 300001, 4000, 64, 65, 123457, 200000, 9, 50000, 777, 150000][k]
            xi, xq = slicer_input(n, 40 + k, "smooth"), slicer_input(n, 90 + k, "noise")
            if k % 4 == 1:
                s = QuadratureSlicer(sample_rate=48000, config="qpsk_2400")
                iq = IQData()
                iq.i_data, iq.q_data = xi, xq
                inputs.append(iq)
            else:
                s = BinarySlicer(sample_rate=48000, config="9600" if k % 2 else "1200")
                if k == 7:
                    s.retune(symbol_rate=5.0)            # 9600 samples per symbol: a byte every 76800 samples
                    xi = np.where((np.arange(n) // 31000) % 2 == 0, 1.0, -1.0)      # few crossings, or the clock never gets there
                inputs.append(xi)
            slicers.append(s)
        return slicers, [s.sign_bitmaps(x) for s, x in zip(slicers, inputs)]

    slicers, bitmaps = build()
    plain = slice_batch(slicers, bitmaps)
    slicers2, bitmaps2 = build()
    fetch = slice_batch(slicers2, bitmaps2, ctx, defer=True, compact=True, reserve=2.0)
    compact = fetch(ctx)
    assert sum(len(p) for p in plain) > 3000
    # stream 7 needs more lockstep launches than a fresh context's first guess: the batch is emitted twice, and the partial byte left
    # for the next call must be the final trajectory's (it used to keep bits of the first emission)
    assert slicers[7]._state.working_bits == 5 and slicers[7]._state.working_byte == 3
    # the native host stage reads the compact form as it is (before anything below asks for .address, which expands it)
    chain = lambda: ["c", None, None, L.LFSR(poly=0x3, invert=True), cb.CodecConfigurator({"type": "ax25"}, "c")]
    assert compact[2].address_steps is not None and len(compact[2]) > 900
    rows_b = ce._host_rows([chain()], [compact[2]])
    assert compact[2].address_steps is not None
    rows_a = ce._host_rows([chain()], [plain[2]])
    assert rows_a[0].tobytes() == rows_b[0].tobytes()
    for k, (a, b) in enumerate(zip(plain, compact)):
        assert np.array_equal(a.data, b.data), k
        if k == 7:
            assert b.address_steps is None and len(b) >= 2, (len(b), b.address_steps, a.address)      # the wide steps were noticed
        elif len(b):
            assert b.address_steps is not None, k
        assert np.array_equal(a.address, b.address), k
        sa, sb = slicers[k]._state, slicers2[k]._state
        for f in ("phase_clock", "last_i_negative", "last_q_negative", "working_byte", "working_bits", "state_register", "streamaddress"):
            assert getattr(sa, f) == getattr(sb, f), (k, f, getattr(sa, f), getattr(sb, f))


def test_slice_batch_rejects_bad_jobs(ctx):
    from pymodem_amd import NativeError
    from pymodem_amd._native import SliceJob
    bits = ctx.upload(np.zeros(4, np.uint64))
    out = ctx.empty(64, np.uint8)
    addr = ctx.empty(64, np.int64)
    def job(**kw):
        j = (SliceJob * 1)()
        j[0].d_bits_i, j[0].n, j[0].d_data, j[0].d_addr, j[0].cap = bits.ptr, 100, out.ptr, addr.ptr, 64
        j[0].params.samples_per_symbol, j[0].params.lock_rate, j[0].params.bits_per_symbol, j[0].params.state_mask = 40.0, 0.75, 1, 3
        for k, v in kw.items():
            setattr(j[0].params, k, v)
        return j
    chk(L().pm_slice_batch(ctx.handle, job(), 1))
    for bad in (dict(bits_per_symbol=3), dict(samples_per_symbol=0.0), dict(lock_rate=float("nan"))):
        with pytest.raises(NativeError):
            chk(L().pm_slice_batch(ctx.handle, job(**bad), 1))
    with pytest.raises(NativeError):
        chk(L().pm_slice_batch(ctx.handle, job(), 65))


def test_fir_signs_batch_matches_single_launches(ctx):
    """pm_fir_signs_f64_batch: several streams of different lengths, one launch; each bitmap equals its own pm_fir_signs_f64."""
    rng = np.random.default_rng(64)
    m = 100
    h = rng.standard_normal(m)
    dh = ctx.upload(h)
    lens = [m, m + 1, 5000, 70003, 69983, 12345, 4096 + m - 1, 300000, 299981]
    xs = [rng.standard_normal(n) for n in lens]
    dx = [ctx.upload(x) for x in xs]
    bits = [ctx.empty((n - m + 1 + 63) // 64 + 1, np.uint64) for n in lens]
    g = len(lens)
    px, pn, pb = (ctypes.c_void_p * g)(), (ctypes.c_int64 * g)(), (ctypes.c_void_p * g)()
    for j in range(g):
        px[j], pn[j], pb[j] = dx[j].ptr.value, lens[j], bits[j].ptr.value
    for flags in (0, 1):
        chk(L().pm_fir_signs_f64_batch(ctx.handle, g, px, pn, dh.ptr, m, pb, flags))
        for j in range(g):
            nout = lens[j] - m + 1
            got = np.unpackbits(bits[j].download().view(np.uint8), bitorder="little")[:nout].astype(bool)
            y = O.fir_canon(xs[j], h)
            assert np.array_equal(got, (-y if flags else y) >= 0), (j, flags)
    from pymodem_amd import NativeError
    with pytest.raises(NativeError):
        chk(L().pm_fir_signs_f64_batch(ctx.handle, 17, px, pn, dh.ptr, m, pb, 0))
    pn[0] = m - 1
    with pytest.raises(NativeError):
        chk(L().pm_fir_signs_f64_batch(ctx.handle, g, px, pn, dh.ptr, m, pb, 0))


def _exact_afsk_signs(ctx, x, mark, space_pair, lpf):
    n, m, ml = len(x), len(mark[0]), len(lpf)
    dx = ctx.upload(x)
    t = [ctx.upload(v) for v in (mark[0], mark[1], space_pair[0], space_pair[1])]
    c = ctx.empty(n - m + 1, np.float64)
    chk(L().pm_afsk_correlate(ctx.handle, dx.ptr, n, t[0].ptr, t[1].ptr, t[2].ptr, t[3].ptr, m, c.ptr))
    nout = n - m - ml + 2
    bits = ctx.empty((nout + 63) // 64 + 1, np.uint64)
    dl = ctx.upload(lpf)
    chk(L().pm_fir_signs_f64(ctx.handle, c.ptr, c.n, dl.ptr, ml, bits.ptr, 0))
    return np.unpackbits(bits.download().view(np.uint8), bitorder="little")[:nout].astype(bool)


def _tones(mark, unit):
    from pymodem_amd import taps as T
    from pymodem_amd._native import AfskTones
    mk, sp = T.tone_model(*mark), T.tone_model(*unit)
    tones = AfskTones()
    tones.mark_rot[:], tones.mark_end[:], tones.space_rot[:], tones.space_end[:] = mk[0], mk[1], sp[0], sp[1]
    tones.tap_dev = max(mk[2], sp[2])
    return tones


def _sweep(ctx, x, x_bound, mark, unit, gains, lpf, sliding=False):
    n, m, ml, g = len(x), len(mark[0]), len(lpf), len(gains)
    nout = n - m - ml + 2
    space = np.stack([np.stack([gn * unit[0], gn * unit[1]]) for gn in gains])
    dx, dl = ctx.upload(x), ctx.upload(lpf)
    t = [ctx.upload(v) for v in (mark[0], mark[1], unit[0], unit[1])]
    ds = ctx.upload(space.reshape(-1))
    bits = [ctx.empty((nout + 63) // 64 + 1, np.uint64) for _ in range(g)]
    ptrs = (ctypes.c_void_p * g)(*[b.ptr.value for b in bits])
    gs = (ctypes.c_double * g)(*gains)
    redo = ctypes.c_int64()
    args = (ctx.handle, dx.ptr, n, float(x_bound), t[0].ptr, t[1].ptr, t[2].ptr, t[3].ptr, ds.ptr, gs, g, m, dl.ptr, ml,
            float(np.abs(lpf).sum()), ptrs)
    if sliding:
        chk(L().pm_afsk_sweep_signs_tones(*args, ctypes.byref(_tones(mark, unit))))
    else:
        chk(L().pm_afsk_sweep_signs(*args))
    chk(L().pm_afsk_sweep_last(ctx.handle, ctypes.byref(redo)))
    out = [np.unpackbits(b.download().view(np.uint8), bitorder="little")[:nout].astype(bool) for b in bits]
    return out, redo.value, space


@pytest.mark.parametrize("sliding", [False, True, "unfused", "lpf8", "lpf8-list"])
def test_afsk_gain_sweep_signs_are_the_exact_chain_s(ctx, sliding):
    """pm_afsk_sweep_signs / pm_afsk_sweep_signs_tones (sliding correlator sums, fused with the low-passes or not): bitmaps of a space_gain sweep from ONE unit space correlator pair and two low-passes, certified against
    the exact chain -- every bit must equal pm_afsk_correlate + pm_fir_signs_f64 with that modem's own (gain-scaled) taps, on an
    AFSK-like signal, on noise at several amplitudes (the smaller the amplitude against the caller's bound, the more samples are
    recomputed exactly, and past 65536 of them the exact chains of all modems run instead, decided on the device)."""
    # "unfused": sliding sums, low-passes and combine as three kernels (the path long filters fall back to); "lpf8": the fused kernel
    # with its low-passes as int8 digit products on the matrix pipe (what pm_pipe_* runs)
    # ("lpf8": its uncertain samples decided by the workgroup that found them; "lpf8-list": all of them through the list and
    # sweep_exact_kernel, as in round 4 and as a workgroup's overflow still goes)
    with tuned(ctx, afsk_unfused=int(sliding == "unfused"), afsk_lpf8=int(sliding in ("lpf8", "lpf8-list")), sweep_no_tail=int(sliding == "lpf8-list")):
        _gain_sweep_cases(ctx, sliding)


def _gain_sweep_cases(ctx, sliding):
    from pymodem_amd import taps as T
    rng = np.random.default_rng(1200)
    mi, mq, ui, uq = T.afsk_tone_correlators(48000.0, 1200.0, 1300.0, 2100.0, 1.0, 1.5, 0.0)
    lpf = T.windowed_sinc(100, 900.0, 48000.0, pass_zero=True)
    gains = [1.25, 1.5, 1.75, 2.0, 2.25, 2.5, 2.75]
    n = 300000
    t = np.arange(n)
    tone = np.where((t // 40) % 3 == 0, 1300.0, 2100.0)
    afsk = 8000.0 * np.sin(2 * np.pi * np.cumsum(tone) / 48000.0) + 800.0 * rng.standard_normal(n)
    cases = [("afsk", afsk, 4.0e4), ("noise", 3000.0 * rng.standard_normal(n), 4.0e4), ("quiet noise", 3.0 * rng.standard_normal(n), 4.0e4),
             ("very quiet", 1e-3 * rng.standard_normal(n), 4.0e4)]
    took_exact_chains = []
    for name, x, bound in cases:
        got, redo, space = _sweep(ctx, x, bound, (mi, mq), (ui, uq), gains, lpf, sliding)
        if redo > 65536:
            took_exact_chains.append(name)                    # too many uncertain samples: the gated exact chains wrote the bitmaps
        for g, gn in enumerate(gains):
            assert np.array_equal(gn * ui, space[g, 0])
            want = _exact_afsk_signs(ctx, x, (mi, mq), (space[g, 0], space[g, 1]), lpf)
            assert np.array_equal(got[g], want), (name, gn, int(np.count_nonzero(got[g] != want)), redo)
    assert "afsk" not in took_exact_chains and "noise" not in took_exact_chains and "very quiet" in took_exact_chains
    z = np.zeros(150000)
    got, redo, space = _sweep(ctx, z, 4.0e4, (mi, mq), (ui, uq), gains, lpf, sliding)
    assert redo > 65536                                       # every output is exactly zero: nothing can be certified ...
    for g in range(len(gains)):                               # ... and the exact chains say ">= 0" everywhere
        assert got[g].all()


@pytest.mark.parametrize("rate,baud,mark,space,span", [(48000.0, 1200.0, 1300.0, 2100.0, 1.5), (48000.0, 1200.0, 1600.0, 1800.0, 1.0),
                                                       (8000.0, 300.0, 1600.0, 1800.0, 1.0), (44100.0, 1200.0, 1200.0, 2200.0, 1.5)])
def test_sliding_correlator_sums_stay_within_their_bound(ctx, rate, baud, mark, space, span):
    """pm_afsk_magnitudes: the sliding sums (Z(k+1) = x[k+m] + r Z(k) - r^m x[k], restarted every 16 outputs) against the direct sums
    in the reference's order -- equal at every restart, elsewhere within the bound the certified decision adds for them, and in
    practice orders of magnitude inside it (the bound is what parity rests on; this pins the error model to the hardware)."""
    from pymodem_amd import taps as T
    mi, mq, ui, uq = T.afsk_tone_correlators(rate, baud, mark, space, 1.0, span, 0.0)
    m = len(mi)
    rng = np.random.default_rng(int(rate + mark))
    tones = _tones((mi, mq), (ui, uq))
    assert tones.tap_dev < 1e-13
    dt = [ctx.upload(v) for v in (mi, mq, ui, uq)]
    for n, amp in [(m, 1.0), (m + 15, 3.0e4), (m + 16, 3.0e4), (m + 2047, 1.0e-3), (m + 2048, 3.0e4), (200001, 3.0e4)]:
        x = amp * rng.standard_normal(n)
        x[n // 2:n // 2 + 3 * m] = amp * 4.0                                     # a flat stretch: long cancellation in the sums
        bound_x = float(np.abs(x).max())
        dx = ctx.upload(x)
        nc = n - m + 1
        out = [ctx.empty(nc, np.float64) for _ in range(4)]
        e = ctypes.c_double(-1.0)
        chk(L().pm_afsk_magnitudes(ctx.handle, dx.ptr, n, bound_x, dt[0].ptr, dt[1].ptr, dt[2].ptr, dt[3].ptr, m, None, out[0].ptr, out[1].ptr,
                                   ctypes.byref(e)))
        assert e.value == 0.0
        chk(L().pm_afsk_magnitudes(ctx.handle, dx.ptr, n, bound_x, dt[0].ptr, dt[1].ptr, dt[2].ptr, dt[3].ptr, m, ctypes.byref(tones),
                                   out[2].ptr, out[3].ptr, ctypes.byref(e)))
        M, S, Ms, Ss = [o.download() for o in out]
        # the direct sums are the reference's: mark - space is what pm_afsk_correlate gives
        assert np.array_equal(M - S, O.afsk_correlate_canon(x, mi, mq, ui, uq))
        # every run starts from the direct sums: there only the root differs (not the IEEE one: a reciprocal-square-root seed and ONE
        # Newton step, relative error below 1.5e-12 -- slide_sqrt; in practice ~1e-15)
        assert np.abs(M[::16] - Ms[::16]).max() <= 1.6e-12 * np.abs(M).max() and np.abs(S[::16] - Ss[::16]).max() <= 1.6e-12 * np.abs(S).max()
        worst = max(np.abs(M - Ms).max(), np.abs(S - Ss).max())
        assert 0.0 < e.value < 1e-9 * m * bound_x
        assert worst <= e.value / 8.0, (n, amp, worst, e.value)


@pytest.mark.parametrize("fused", [True, False, "lpf8"])
@pytest.mark.parametrize("gain", [1.0, 2.25])
@pytest.mark.parametrize("rate,baud,mark,space,span", [(48000.0, 1200.0, 1600.0, 1800.0, 1.0), (8000.0, 300.0, 1600.0, 1800.0, 1.0)])
def test_one_chain_certified_signs_are_the_exact_chain_s(ctx, rate, baud, mark, space, span, gain, fused):
    """pm_afsk_sweep_signs_tones with ONE modem: mark - gain * space from the sliding sums as one stream, one low-pass, certified;
    every bit equals pm_afsk_correlate + pm_fir_signs_f64 (signal, loud and quiet noise, silence -> the gated exact chain)."""
    with tuned(ctx, afsk_unfused=int(not fused), afsk_lpf8=int(fused == "lpf8")):
        _one_chain_cases(ctx, rate, baud, mark, space, span, gain)


def _one_chain_cases(ctx, rate, baud, mark, space, span, gain):
    from pymodem_amd import taps as T
    rng = np.random.default_rng(int(rate + 10 * gain))
    mi, mq, ui, uq = T.afsk_tone_correlators(rate, baud, mark, space, 1.0, span, 0.0)
    lpf = T.windowed_sinc(round(rate * 2.5 / baud) | 1, 0.75 * baud, rate, pass_zero=True)
    n = 250001
    t = np.arange(n)
    tone = np.where((t // int(rate / baud)) % 3 == 0, mark, space)
    sig = 8000.0 * np.sin(2 * np.pi * np.cumsum(tone) / rate) + 800.0 * rng.standard_normal(n)
    for name, x in [("signal", sig), ("noise", 3000.0 * rng.standard_normal(n)), ("quiet", 1e-3 * rng.standard_normal(n)), ("silence", np.zeros(n))]:
        got, redo, space_taps = _sweep(ctx, x, 4.0e4, (mi, mq), (ui, uq), [gain], lpf, sliding=True)
        want = _exact_afsk_signs(ctx, x, (mi, mq), (space_taps[0, 0], space_taps[0, 1]), lpf)
        assert np.array_equal(got[0], want), (name, int(np.count_nonzero(got[0] != want)), redo)
        assert redo > 65536 if name == "silence" else (redo <= 65536 or name == "quiet"), (name, redo)


def test_refused_sweep_leaves_no_stale_counter_behind(ctx):
    """The counters of uncertain samples live in a ring of 64; a sweep's last launch clears the NEXT slot.  A call that took its slot
    and was then refused launched nothing, and the sweep after it started from what its slot held 64 sweeps earlier -- up to 65536
    stale list entries handed to the exact recomputation (found as a GPU fault when the test order changed).  Refusals are now
    decided before the ring moves, and a sweep that fails later clears the next slot by hand: every third sweep here is refused,
    every third leaves a large count, and the ones in between must stay exact and report their own count, all the way round the ring."""
    from pymodem_amd import NativeError
    from pymodem_amd import taps as T
    mi, mq, ui, uq = T.afsk_tone_correlators(48000.0, 1200.0, 1300.0, 2100.0, 1.0, 1.5, 0.0)
    lpf = T.windowed_sinc(100, 900.0, 48000.0, pass_zero=True)
    rng = np.random.default_rng(64)
    x = 3000.0 * rng.standard_normal(12000)
    z = np.zeros(12000)
    want = None
    bad = _tones((mi, mq), (ui, uq))
    bad.tap_dev = 1e-3
    for it in range(70):
        got, redo, space = _sweep(ctx, z, 4.0e4, (mi, mq), (ui, uq), [1.5], lpf, sliding=True)          # nothing certifiable: a large count
        assert redo > 10000 and got[0].all()
        dx, dl = ctx.upload(x), ctx.upload(lpf)
        t = [ctx.upload(v) for v in (mi, mq, ui, uq)]
        ds = ctx.upload(np.stack([1.5 * ui, 1.5 * uq]).reshape(-1))
        bits = ctx.empty(len(x) // 64 + 2, np.uint64)
        args = (ctx.handle, dx.ptr, len(x), 4.0e4, t[0].ptr, t[1].ptr, t[2].ptr, t[3].ptr, ds.ptr, (ctypes.c_double * 1)(1.5), 1, len(mi), dl.ptr,
                len(lpf), float(np.abs(lpf).sum()), (ctypes.c_void_p * 1)(bits.ptr.value))
        with pytest.raises(NativeError):
            chk(L().pm_afsk_sweep_signs_tones(*args, ctypes.byref(bad)))                                  # refused
        got, redo, space = _sweep(ctx, x, 4.0e4, (mi, mq), (ui, uq), [1.5], lpf, sliding=True)
        if want is None:
            want = _exact_afsk_signs(ctx, x, (mi, mq), (space[0, 0], space[0, 1]), lpf)
        assert redo < 100 and np.array_equal(got[0], want), (it, redo)


def test_sweep_tones_rejects_templates_that_are_not_tones(ctx):
    from pymodem_amd import NativeError
    from pymodem_amd import taps as T
    mi, mq, ui, uq = T.afsk_tone_correlators(48000.0, 1200.0, 1300.0, 2100.0, 1.0, 1.5, 0.0)
    lpf = T.windowed_sinc(100, 900.0, 48000.0, pass_zero=True)
    x = np.random.default_rng(2).standard_normal(20000) * 100.0
    dx, dl = ctx.upload(x), ctx.upload(lpf)
    t = [ctx.upload(v) for v in (mi, mq, ui, uq)]
    ds = ctx.upload(np.stack([1.5 * ui, 1.5 * uq]).reshape(-1))
    bits = ctx.empty(len(x) // 64 + 2, np.uint64)
    ptrs = (ctypes.c_void_p * 1)(bits.ptr.value)
    gs = (ctypes.c_double * 1)(1.5)
    args = (ctx.handle, dx.ptr, len(x), 500.0, t[0].ptr, t[1].ptr, t[2].ptr, t[3].ptr, ds.ptr, gs, 1, len(mi), dl.ptr, len(lpf), float(np.abs(lpf).sum()), ptrs)
    tones = _tones((mi, mq), (ui, uq))
    tones.tap_dev = 1e-3
    with pytest.raises(NativeError):
        chk(L().pm_afsk_sweep_signs_tones(*args, ctypes.byref(tones)))
    with pytest.raises(NativeError):
        chk(L().pm_afsk_sweep_signs_tones(*args, None))
    tones.tap_dev = 2e-15
    chk(L().pm_afsk_sweep_signs_tones(*args, ctypes.byref(tones)))             # and the same call with honest tones goes through


def test_runtime_additions(ctx):
    """pm_d2d, pm_host_pin, pm_event_query / pm_event_sync, pm_ctx_scratch and a CU-masked context (pm_ctx_create_cumask) do what they say."""
    import pymodem_amd
    from pymodem_amd._native import check, lib
    L = lib()
    a = np.arange(100000, dtype=np.int16)
    src, dst = ctx.upload(a), ctx.empty(len(a), np.int16)
    check(L.pm_d2d(ctx.handle, dst.ptr, src.ptr, a.nbytes))
    ev = ctx.record_event()
    pymodem_amd.Context.event_sync(ev)
    assert pymodem_amd.Context.event_done(ev) is True
    assert np.array_equal(dst.download(), a)
    # a page-locked host block takes the same copy; the executor's recycled result blocks are pinned when they are made
    blk = np.zeros(a.nbytes, np.uint8)
    check(L.pm_host_pin(ctx.handle, blk.ctypes.data_as(ctypes.c_void_p), blk.nbytes))
    check(L.pm_d2h(ctx.handle, blk.ctypes.data_as(ctypes.c_void_p), dst.ptr, blk.nbytes))
    check(L.pm_host_unpin(blk.ctypes.data_as(ctypes.c_void_p)))
    assert np.array_equal(blk.view(np.int16), a)
    assert np.array_equal(dst.download(recycle=True), a) and np.array_equal(dst.download(recycle=True), a)
    have = ctypes.c_size_t()
    check(L.pm_ctx_scratch(ctx.handle, 0, ctypes.byref(have)))
    check(L.pm_ctx_scratch(ctx.handle, have.value + (1 << 20), ctypes.byref(have)))
    assert have.value >= (1 << 20)
    # a context confined to the first 4 CUs of every XCD runs the same kernels with the same results
    cus = L.pm_device_cus(0)
    assert cus >= 64
    low, high = pymodem_amd.Context.cu_split(0, 4)
    assert sum(bin(w).count("1") for w in low) == 32 and sum(bin(w).count("1") for w in low + high) == cus
    masked = pymodem_amd.Context(0, cu_mask=low)
    x = noise_i16(200000)
    h = np.random.default_rng(5).standard_normal(40)
    y = masked.empty(len(x) - 39, np.float64)
    check(L.pm_fir_valid_i16(masked.handle, masked.upload(x).ptr, len(x), masked.upload(h).ptr, 40, y.ptr, 0))
    assert np.array_equal(y.download(), O.fir_canon(x, h))
    masked.close()


def test_afsk_group_run_equals_the_separate_calls(ctx, config_lines):
    """pm_afsk_group_run (band-pass + every certified sweep of a chain group in one call, what the group executor's fast path uses)
    leaves the bitmaps of modem.front_end() + AFSKModem.sweep_signs() per sweep."""
    from pymodem_amd import chain_builder as cb, chain_execute as ce, siggen
    lines = config_lines("afsk_1200_ax25_super_opt.json")
    audio, _ = siggen.recording("afsk1200_ax25", 48000, packets=6, seed=33, noise_sigma=1500.0, payload_len=(20, 60))
    res = {}
    for fast in (True, False):
        ce._USE_GROUP_NATIVE = fast
        try:
            st = {}
            ce.process_chains_device([cb.build_chain(48000, l) for l in lines], audio, stages=st)
        finally:
            ce._USE_GROUP_NATIVE = True
        res[fast] = [(np.array(st["sliced"][c].data), np.array(st["sliced"][c].address)) for c in range(len(lines))]
    for c in range(len(lines)):
        assert len(res[True][c][0]) > 100
        assert np.array_equal(res[True][c][0], res[False][c][0]) and np.array_equal(res[True][c][1], res[False][c][1]), c


def test_profiler_intervals_share_a_time_base_across_contexts(ctx):
    """pm_prof_intervals: every tracked launch's (start, end) on the device's common time base -- launches of one context are
    disjoint and in order, launches on a second context fall in the same time span, and the sums equal pm_prof_read's."""
    import pymodem_amd
    side = pymodem_amd.Context.side(index=78, high_priority=False)
    n, m = 2_000_000, 100
    taps = ctx.upload(np.linspace(-1.0, 1.0, m))
    x = ctx.upload(np.random.default_rng(2).standard_normal(n))
    y1, y2 = ctx.empty(n - m + 1, np.float64), side.empty(n - m + 1, np.float64)
    ctx.sync()
    for c in (ctx, side):
        c.profile(True)
    for _ in range(5):
        chk(L().pm_fir_valid_f64(ctx.handle, x.ptr, n, taps.ptr, m, y1.ptr, 0))
        chk(L().pm_fir_valid_f64(side.handle, x.ptr, n, taps.ptr, m, y2.ptr, 0))
    ctx.sync()
    side.sync()
    iv = {name: c.profile_intervals("fir_f64") for name, c in (("ctx", ctx), ("side", side))}
    read = {name: c.profile_read()["fir_f64"] for name, c in (("ctx", ctx), ("side", side))}
    for c in (ctx, side):
        c.profile(False)
    for name, v in iv.items():
        assert v.shape == (5, 2) and np.all(v[:, 1] > v[:, 0]) and np.all(v[1:, 0] >= v[:-1, 1] - 1e-3), name      # disjoint, in order
        assert abs(float(np.sum(v[:, 1] - v[:, 0])) - read[name][0]) < 1e-2 and read[name][1] == 5
    lo, hi = min(v[:, 0].min() for v in iv.values()), max(v[:, 1].max() for v in iv.values())
    assert hi - lo < 1000.0 and iv["side"][:, 0].min() < iv["ctx"][:, 1].max()        # one time base: the two streams' launches interleave
    assert ctx.profile_intervals("fir_f64").shape == (0, 2)                            # switching off clears
