"""Oracle primitives vs the reference (tests/golden/primitives.npz), bit for bit.
AGC agc.py:26-80 - NCO nco.py:34-53 - IIR iir.py:38-54 - PI pi_control.py:25-33 - phase detector
phase_detector.py:12-45,124-149 - LFSR lfsr.py:22-52 - CRC crc_functions.py - GF/RS gf_functions.py,
rs_functions.py:33-150 - slicers slicer.py:59-107,193-242."""
import ctypes

import numpy as np
import pytest

from oracle import oracle as O


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a, b)


def test_agc(golden):
    g = golden("primitives")
    for name, rate in [("8k", 8000.0), ("48k", 48000.0)]:
        buf = g[f"agc_{name}_in"].copy()
        out, env = O.agc_apply(buf, rate, 500.0, 0.5, 50.0, 1.0, want_env=True)
        assert same(out, g[f"agc_{name}_out"]) and same(env, g[f"agc_{name}_env"])
    for name in ["neg", "zero"]:           # normal < 0; leading zeros (envelope == 0 leaves samples untouched)
        buf = g[f"agc_{name}_in"].copy()
        out, _ = O.agc_apply(buf, 8000.0, 500.0, 0.01, 50.0, 1.0)
        assert same(out, g[f"agc_{name}_out"])


def test_nco_iir_pi(golden):
    g = golden("primitives")
    tab = O.nco_table()
    assert same(tab, g["nco_table"])
    L = O.make_loop(48000.0, 1500.0, 250.0, 1.0, 0, 0, 0, 0)
    ctl = g["nco_ctl"]
    n = len(ctl)
    s, c, ph = np.empty(n), np.empty(n), np.empty(n)
    O.lib().pmo_nco_run(ctypes.byref(L), O._p(tab), O._p(ctl), ctypes.c_int64(n), O._p(s), O._p(c), O._p(ph))
    assert same(s, g["nco_sin"]) and same(c, g["nco_cos"]) and same(ph, g["nco_phase"])
    x = g["iir_in"]
    for rate, fc, gn, name in [(48000.0, 250.0, 1.0, "a"), (8000.0, 150.0, 2.0, "b")]:
        L = O.make_loop(rate, 0, fc, gn, 0, 0, 0, 0)
        assert same(np.array([L.b0, L.b1, L.a1]), g[f"iir_{name}_coefs"])
        y = np.empty(len(x))
        O.lib().pmo_iir_run(ctypes.byref(L), O._p(x), ctypes.c_int64(len(x)), O._p(y))
        assert same(y, g[f"iir_{name}_out"])
    L = O.make_loop(48000.0, 0, 250.0, 1.0, 0.06, 0.06 / 1000, 31.25, 7200)
    x = g["pi_in"]
    y, ig = np.empty(len(x)), np.empty(len(x))
    O.lib().pmo_pi_run(ctypes.byref(L), O._p(x), ctypes.c_int64(len(x)), O._p(y), O._p(ig))
    assert same(y, g["pi_out"]) and same(ig, g["pi_integral"])


def test_phase_detector(golden):
    g = golden("primitives")
    assert same(O.pd_table().astype(np.int64), g["pd_table"])
    assert same(O.pd_lookup(g["pd_re"], g["pd_im"]).astype(np.int64), g["pd_err"])


def test_lfsr_crc(golden):
    g = golden("primitives")
    for poly, inv, name in [(0x1, False, "p1"), (0x3, True, "p3i"), (0x63003, True, "g3ruh"), (0x3, False, "p3"), (0x1, True, "p1i")]:
        assert same(O.LFSR(poly, inv).stream_unscramble_8bit(g["lfsr_in"]), g[f"lfsr_{name}_out"])
    flat, pos = g["crc_msgs_flat"], 0
    for ln, cv in zip(g["crc_msgs_len"], g["crc_values"]):
        assert O.crc16(flat[pos:pos + ln]) == cv
        pos += ln


def test_gf_rs(golden):
    g = golden("primitives")
    gf = O.gf256()
    assert same(gf.table, g["gf_table"]) and same(gf.index, g["gf_index"]) and same(gf.inverse, g["gf_inverse"])
    rs = {2: O.RS(0, 2), 16: O.RS(0, 16)}
    for r in rs:
        assert same(rs[r].genpoly, g[f"rs_genpoly_{r}"])
    rets = g["rs_case_ret"]
    assert (rets < 0).any() and (rets > 0).any()          # failures and corrections are both exercised
    for i in range(len(rets)):
        buf = [int(v) for v in g["rs_case_in"][i]]
        ret = rs[int(g["rs_case_roots"][i])].decode(buf, int(g["rs_case_len"][i]), int(g["rs_case_mindist"][i]))
        assert ret == rets[i] and buf == [int(v) for v in g["rs_case_out"][i]], i


BIN = [(48000, "1200", "0.77", "b1200_48k"), (8000, "300", "0.90", "b300_8k"), (44100, "1200", "0.75", "b1200_44k"),
       (48000, "9600", "0.88", "b9600_48k"), (22050, "9600", "0.88", "b9600_22k")]
QUAD = [(48000, "qpsk_2400", "0.98", "q2400_48k"), (8000, "qpsk_600", "0.815", "q600_8k"), (48000, "bpsk_1200", "0.9", "qb1200_48k"),
        (44100, "qpsk_3600", "0.985", "q3600_44k"), (48000, "bpsk_300", "0.815", "qb300_48k"), (48000, "qpsk_4800", "0.99", "q4800_48k")]


@pytest.mark.parametrize("rate,cfg,lock,name", BIN)
def test_binary_slicer(golden, rate, cfg, lock, name):
    g = golden("primitives")
    s = O.BinarySlicer(rate, cfg, {"lock_rate": lock})
    d, a = s.slice(g["slicer_in"])
    assert same(d, g[f"slicer_{name}_data"]) and same(a, g[f"slicer_{name}_addr"])
    assert s.state[0] == float(g[f"slicer_{name}_clk"])


@pytest.mark.parametrize("rate,cfg,lock,name", QUAD)
def test_quadrature_slicer(golden, rate, cfg, lock, name):
    g = golden("primitives")
    s = O.QuadratureSlicer(rate, cfg, {"lock_rate": lock})
    d, a = s.slice((g["slicer_in"], g["slicer_in_q"]))
    assert same(d, g[f"slicer_{name}_data"]) and same(a, g[f"slicer_{name}_addr"])
    assert s.state[0] == float(g[f"slicer_{name}_clk"])
