"""Oracle tap designers vs the reference's taps (tests/golden/taps.npz), bit for bit.
Reference: afsk.py:102-146, fsk.py:115-138, psk.py:111-160,639-703, afsk_pll.py:84-138, rrc.py:18-95,
hilbert.py:9-34; scipy.signal.firwin restated in oracle.firwin_hamming."""
import numpy as np
import pytest

from oracle import oracle as O

RATES = [8000, 11025, 22050, 44100, 48000, 96000]


def same(a, b):
    return a.shape == b.shape and np.array_equal(a, b)


@pytest.mark.parametrize("rate", RATES)
def test_modem_taps_bit_exact(golden, rate):
    g = golden("taps")
    for cfg in ["300", "1200"]:
        m = O.AFSKModem(rate, cfg, {})
        for k, v in [("bpf", m.input_bpf), ("lpf", m.output_lpf), ("mi", m.mi), ("mq", m.mq), ("si", m.si), ("sq", m.sq)]:
            assert same(g[f"afsk_{cfg}_{rate}_{k}"], v), (cfg, k)
    m = O.AFSKModem(rate, "1200", {"space_gain": "1.75", "mark_freq": "1300.0", "space_freq": "2100.0", "correlator_span": "1.5"})
    for k, v in [("mi", m.mi), ("mq", m.mq), ("si", m.si), ("sq", m.sq)]:
        assert same(g[f"afsk_1200opt_{rate}_{k}"], v), k
    for cfg in ["300", "1200"]:
        m = O.BPSKModem(rate, cfg, {})
        assert same(g[f"bpsk_{cfg}_{rate}_bpf"], m.input_bpf) and same(g[f"bpsk_{cfg}_{rate}_rrc"], m.rrc)
    for cfg in ["qpsk_3600", "qpsk_600", "qpsk_2400", "bpsk_300", "bpsk_1200"]:
        m = O.MPSKModem(rate, cfg, {})
        assert same(g[f"mpsk_{cfg}_{rate}_bpf"], m.input_bpf)
        assert same(g[f"mpsk_{cfg}_{rate}_hilbert"], m.hilbert)
        assert same(g[f"mpsk_{cfg}_{rate}_rrc"], m.rrc)
        assert m.delay == int(g[f"mpsk_{cfg}_{rate}_delay"])
    m = O.AFSKPLLModem(rate, "300", {})
    assert same(g[f"pll_300_{rate}_bpf"], m.input_bpf) and same(g[f"pll_300_{rate}_lpf"], m.output_lpf)
    for cfg in ["9600", "4800", "4800-rrc", "9600-rrc", "4800-gauss", "9600-gauss"]:
        key = f"fsk_{cfg}_{rate}_lpf"
        if key in g.files:
            assert same(g[key], O.FSKModem(rate, cfg, {}).input_lpf), cfg


def test_rrc_windows_and_hilbert(golden):
    g = golden("taps")
    for w in ["rect", "hann", "blackmann", "blackmann-harris", "flattop", "tukey"]:
        assert same(g[f"rrc_window_{w}"], O.rrc_taps(48000, 1200, 6, 0.3, w)), w
    for n in [21, 49, 131, 163, 217]:
        assert same(g[f"hilbert_{n}"], O.hilbert_taps(n)[0])


def test_firwin_restatement_matches_scipy_here():
    """Extra pin: the SciPy in this image agrees with the restatement on shapes the configs never use."""
    ss = pytest.importorskip("scipy.signal")
    for n, cut, fs, pz in [(31, [300.0, 3000.0], 44100, False), (64, [1200.0, 1800.0], 8000, False),
                           (9, 6000.0, 48000, True), (101, [100.0], 8000, True), (7, [1000.0, 2000.0, 3000.0], 12000, True)]:
        ref = ss.firwin(n, cut, pass_zero=pz, fs=fs)
        assert np.array_equal(ref, O.firwin_hamming(n, cut, fs, pz))
