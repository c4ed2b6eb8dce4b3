"""pm_fir8_rows_signs_f64 (csrc/pm_fir8.hip): the long matched filters of the carrier-loop modems (psk.py:193 RRC 961, psk.py:750-751
RRC 241) as certified signs on the int8 matrix pipe.  Its bitmap must be pm_fir_rows_signs_f64's -- the canonical binary64 sum's
`>= 0`, itself bit-exact against the oracle (tests/test_gpu_kernels.py) -- bit for bit, for every input: the matrix pipe only
decides what its proven bound lets it decide, everything else takes the canonical chain."""
import ctypes

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import pymodem_amd
    return pymodem_amd.Context.default()


def _bits(buf, stride, rows, nout):
    w = buf.download().reshape(rows, stride)
    return np.unpackbits(w.view(np.uint8), axis=1, bitorder="little")[:, :nout].astype(bool), w


def _both(ctx, x2d, taps):
    """-> (bits of the matrix-pipe path, bits of the exact rows kernel, outputs recomputed exactly)"""
    from pymodem_amd._native import check, lib
    rows, n = x2d.shape
    m = len(taps)
    nout = n - m + 1
    stride = (nout + 63) // 64 + 3
    dx = ctx.upload(np.ascontiguousarray(x2d).reshape(-1))
    dt = ctx.upload(np.ascontiguousarray(taps, dtype=np.float64))
    b8 = ctx.upload(np.full(rows * stride, 0xA5A5A5A5A5A5A5A5, np.uint64))
    bx = ctx.upload(np.full(rows * stride, 0xA5A5A5A5A5A5A5A5, np.uint64))
    redo = ctypes.c_int64(-1)
    h = np.ascontiguousarray(taps, dtype=np.float64)
    check(lib().pm_fir8_rows_signs_f64(ctx.handle, dx.ptr, n, rows, n, h.ctypes.data_as(ctypes.c_void_p), m, b8.ptr, stride, ctypes.byref(redo)))
    check(lib().pm_fir_rows_signs_f64(ctx.handle, dx.ptr, n, rows, n, dt.ptr, m, bx.ptr, stride, 0))
    got, gw = _bits(b8, stride, rows, nout)
    want, ww = _bits(bx, stride, rows, nout)
    words = (nout + 63) // 64
    assert np.array_equal(gw[:, :words], ww[:, :words])      # whole words: bits past the last output are zero in both
    assert (gw[:, words:] == 0xA5A5A5A5A5A5A5A5).all()       # and nothing is written past the row's words
    return got, want, redo.value


def _taps(m, rng):
    # a matched-filter-like shape: windowed sinc, L2-normalised (rrc.py:48), plus a little asymmetry so that reversal matters
    t = np.arange(m) - (m - 1) / 2
    h = np.sinc(t / (m / 12.0)) * np.hanning(m + 2)[1:-1] + 1e-3 * rng.standard_normal(m)
    return h / np.linalg.norm(h)


@pytest.mark.parametrize("m", [961, 241, 16, 100, 497, 498, 753, 1009])
def test_certified_signs_equal_the_exact_kernel(ctx, m):
    rng = np.random.default_rng(m)
    taps = _taps(m, rng)
    n = 70001 + m
    t = np.arange(n)
    rows = np.stack([
        rng.standard_normal(n),                                                 # noise at unit level
        0.7 * np.sign(np.sin(2 * np.pi * t / 160.0)) * np.cos(0.001 * t) + 0.05 * rng.standard_normal(n),     # a keyed carrier, mixed down
        3.0e4 * rng.standard_normal(n) * (1 + np.sin(t / 5000.0)),              # large and breathing
        1e-6 * rng.standard_normal(n),                                          # small
        np.where((t // 9000) % 2 == 0, rng.standard_normal(n), 0.0),            # stretches of exact zeros (whole windows of them)
        np.round(rng.standard_normal(n) * 4) / 4,                               # coarse values: exact ties at zero do happen
    ])
    got, want, redo = _both(ctx, rows, taps)
    assert np.array_equal(got, want), [int(np.sum(got[r] != want[r])) for r in range(len(rows))]
    total = got.size
    # the matrix pipe decides nearly everything; what it cannot decide by magnitude are sums that ARE zero (windows of zeros)
    silent = sum(int(np.sum(np.convolve((rows[r] != 0).astype(np.int64), np.ones(m, np.int64), "valid") == 0)) for r in range(len(rows)))
    assert 0 <= redo <= silent + 0.02 * total, (redo, silent, total)


def test_certification_is_tight_on_the_modems_own_filters(ctx):
    """The two filters of the bench workloads on AGC'd-noise-like input: how many outputs go to the exact chain (DESIGN.md 4.1c
    quotes these)."""
    from pymodem_amd import chain_builder as cb
    rng = np.random.default_rng(5)
    for kind, cfg in (("bpsk", "300"), ("mpsk", "qpsk_2400")):
        md = cb.ModemConfigurator(48000, {"type": kind, "config": cfg, "options": {}})
        taps = np.asarray(md.rrc_taps, dtype=np.float64)
        n = 400000
        x = (rng.standard_normal((2, n)) * 0.35).clip(-1.2, 1.2)
        got, want, redo = _both(ctx, x, taps)
        assert np.array_equal(got, want)
        assert redo < 2e-3 * got.size, (kind, len(taps), redo, got.size)


def test_against_the_oracle_directly(ctx):
    """One row straight against the CPU restatement's canonical FIR (oracle.fir_canon), not through another HIP kernel."""
    rng = np.random.default_rng(11)
    taps = _taps(241, rng)
    x = rng.standard_normal(30000)
    got, want, _ = _both(ctx, x[None, :], taps)
    y = O.fir_canon(x, taps)
    assert np.array_equal(got[0], y >= 0)


def test_degenerate_inputs(ctx):
    """Silence, denormals, huge values, NaN and infinities: windows the matrix pipe cannot scale are recomputed whole; the result is
    still the exact kernel's."""
    rng = np.random.default_rng(3)
    m = 241
    taps = _taps(m, rng)
    n = 40000
    z = np.zeros(n)
    den = rng.standard_normal(n) * 1e-310
    huge = rng.standard_normal(n) * 1e305
    nan = rng.standard_normal(n)
    nan[12345] = np.nan
    nan[30000] = np.inf
    neg0 = -np.zeros(n)
    mixed = rng.standard_normal(n)
    mixed[:9000] = 0.0
    mixed[20000:20010] = 1e200
    got, want, redo = _both(ctx, np.stack([z, den, huge, nan, neg0, mixed]), taps)
    assert np.array_equal(got, want)
    assert got[0].all() and got[4].all()                     # sums of zeros are +0: `>= 0` holds


def test_short_rows_and_many_rows(ctx):
    rng = np.random.default_rng(9)
    taps = _taps(241, rng)
    for n in (241, 242, 241 + 63, 241 + 64, 241 + 8191, 241 + 8192, 241 + 8193):
        got, want, _ = _both(ctx, rng.standard_normal((3, n)), taps)
        assert np.array_equal(got, want), n
    got, want, _ = _both(ctx, rng.standard_normal((300, 3000)), taps)
    assert np.array_equal(got, want)


def test_more_rows_than_a_launch_s_grid_holds(ctx):
    """The batch engine sends R x C streams through pm_fir8_rows_signs, up to 2^20 of them, and rows are the grid's y dimension (65535 at
    most): 70 000 short rows go in two launches (the call used to refuse them, and the engine had no other path).  Every row of
    the second batch too must be the exact kernel's, nothing written past a row's words."""
    rng = np.random.default_rng(70000)
    m, n, rows = 64, 64 + 191, 70000                          # 192 outputs per row: three bitmap words
    taps = _taps(m, rng)
    x = rng.standard_normal((rows, n)) * np.exp(rng.uniform(-6, 6, (rows, 1)))
    x[::997] = 0.0                                            # some rows of exact zeros
    got, want, _ = _both(ctx, x, taps)
    assert np.array_equal(got, want), int(np.sum(got != want))
    assert np.array_equal(got[65530:65540], want[65530:65540]) and got[-1].any()
