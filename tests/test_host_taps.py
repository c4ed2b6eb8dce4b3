"""Product tap designers (pymodem_amd/taps.py + the presets in pymodem_amd/modems.py, pymodem_amd/slicer.py) against the
reference's taps (tests/golden/taps.npz).  Host logic only."""
import numpy as np
import pytest

RATES = [8000, 11025, 22050, 44100, 48000, 96000]


def close(ref, got):
    """Bit-identical on the NumPy build that made the goldens; a few ulp elsewhere (libm / SIMD sin differ by platform)."""
    return ref.shape == got.shape and (np.array_equal(ref, got) or np.abs(ref - got).max() <= 8 * np.finfo(float).eps * np.abs(ref).max())


@pytest.mark.parametrize("rate", RATES)
def test_modem_presets_and_taps(golden, rate):
    from pymodem_amd import chain_builder as cb
    g = golden("taps")
    for cfg in ["300", "1200"]:
        m = cb.ModemConfigurator(rate, {"type": "afsk", "config": cfg, "options": {}})
        for k, v in [("bpf", m.input_bpf), ("lpf", m.output_lpf), ("mi", m.mark_correlator_i), ("mq", m.mark_correlator_q),
                     ("si", m.space_correlator_i), ("sq", m.space_correlator_q)]:
            assert close(g[f"afsk_{cfg}_{rate}_{k}"], v), (cfg, k)
        assert m.output_sample_rate == 1.0 * rate
    m = cb.ModemConfigurator(rate, {"type": "afsk", "config": "1200", "options": {
        "space_gain": "1.75", "mark_freq": "1300.0", "space_freq": "2100.0", "correlator_span": "1.5", "mark freq": "9"}})
    for k, v in [("mi", m.mark_correlator_i), ("mq", m.mark_correlator_q), ("si", m.space_correlator_i), ("sq", m.space_correlator_q)]:
        assert close(g[f"afsk_1200opt_{rate}_{k}"], v), k
    for cfg in ["300", "1200"]:
        m = cb.ModemConfigurator(rate, {"type": "bpsk", "config": cfg, "options": {}})
        assert close(g[f"bpsk_{cfg}_{rate}_bpf"], m.input_bpf) and close(g[f"bpsk_{cfg}_{rate}_rrc"], m.rrc_taps)
    for cfg in ["qpsk_3600", "qpsk_600", "qpsk_2400", "bpsk_300", "bpsk_1200"]:
        m = cb.ModemConfigurator(rate, {"type": "mpsk", "config": cfg, "options": {}})
        assert close(g[f"mpsk_{cfg}_{rate}_bpf"], m.input_bpf) and close(g[f"mpsk_{cfg}_{rate}_hilbert"], m.hilbert_taps)
        assert close(g[f"mpsk_{cfg}_{rate}_rrc"], m.rrc_taps) and m.hilbert_delay == int(g[f"mpsk_{cfg}_{rate}_delay"])
    m = cb.ModemConfigurator(rate, {"type": "afsk_pll", "config": "300", "options": {}})
    assert close(g[f"pll_300_{rate}_bpf"], m.input_bpf) and close(g[f"pll_300_{rate}_lpf"], m.output_lpf)
    for cfg in ["9600", "4800", "4800-rrc", "9600-rrc", "4800-gauss", "9600-gauss"]:
        key = f"fsk_{cfg}_{rate}_lpf"
        if key in g.files:
            m = cb.ModemConfigurator(rate, {"type": "fsk", "config": cfg, "options": {}})
            assert close(g[key], m.input_lpf), cfg
            assert not hasattr(m, "output_sample_rate")        # fsk.py has none; the runner falls back to the input rate


def test_tables_and_windows(golden):
    from pymodem_amd import taps as T
    g, p = golden("taps"), golden("primitives")
    for w in ["rect", "hann", "blackmann", "blackmann-harris", "flattop", "tukey"]:
        assert close(g[f"rrc_window_{w}"], T.root_raised_cosine(48000, 1200, 6, 0.3, w)), w
    for n in [21, 49, 131, 163, 217]:
        assert close(g[f"hilbert_{n}"], T.hilbert_transformer(n)[0])
    assert np.array_equal(T.sine_wavetable(), p["nco_table"])
    assert np.array_equal(T.qpsk_error_table().astype(np.int64), p["pd_table"])
    assert np.array_equal(np.array(T.one_pole_lowpass(48000.0, 250.0, 1.0)), p["iir_a_coefs"])


def test_factories_mirror_the_reference():
    from pymodem_amd import chain_builder as cb
    assert cb.ModemConfigurator(48000, {"type": "nope", "config": "x", "options": {}}) == []
    assert cb.SlicerConfigurator(48000, {"type": "4level", "config": "x", "options": {}}) == []      # broken upstream, not provided
    assert cb.StreamConfigurator({"type": "other", "options": {}}) == []
    s = cb.StreamConfigurator({"type": "lfsr", "options": {"poly": "0x63003", "invert": "yes"}})
    assert s.polynomial == 0x63003 and s.invert is True
    c = cb.CodecConfigurator({"type": "IL2P", "options": {"crc": "no", "sync_tol": "2", "min_dist": "1"}}, "name")
    assert (c.collect_trailing_crc, c.sync_tolerance, c.min_distance, c.identifier) == (False, 2, 1, "name")
    sl = cb.SlicerConfigurator(48000.0, {"type": "binary", "config": "9600", "options": {"lock_rate": "0.88"}})
    assert sl.samples_per_symbol == 5.0 and sl.rollover_threshold == 2.0 and sl.lock_rate == 0.88
    q = cb.SlicerConfigurator(48000.0, {"type": "quadrature", "config": "qpsk_2400", "options": {"lock_rate": "0.98"}})
    assert (q.state_mask, q.bits_per_symbol, q.symbol_rate) == (0xF, 2, 1200)


def test_tone_model_accepts_tone_templates_only():
    """taps.tone_model: the correlator templates of afsk.py:134-144 are the powers of one rotation to ~1e-15; anything else is far
    from that, and AFSKModem then keeps to the direct correlator sums (pm_afsk_sweep_signs_tones refuses tap_dev >= 1e-6)."""
    import numpy as np
    from pymodem_amd import taps as T
    for rate, baud, mark, space, span in [(48000.0, 1200.0, 1300.0, 2100.0, 1.5), (8000.0, 300.0, 1600.0, 1800.0, 1.0), (44100.0, 1200.0, 1200.0, 2200.0, 1.5)]:
        mi, mq, ui, uq = T.afsk_tone_correlators(rate, baud, mark, space, 1.0, span, 0.0)
        for hi, hq in ((mi, mq), (ui, uq)):
            rot, end, dev = T.tone_model(hi, hq)
            assert dev < 1e-13 and abs(rot[0] ** 2 + rot[1] ** 2 - 1.0) < 1e-15
            m = len(hi)
            w = np.arctan2(rot[1], rot[0])
            assert abs(end[0] - np.cos(m * w)) < 1e-12 and abs(end[1] - np.sin(m * w)) < 1e-12
    rng = np.random.default_rng(0)
    assert T.tone_model(rng.standard_normal(40), rng.standard_normal(40))[2] > 1e-3
    assert T.tone_model(np.ones(1), np.zeros(1)) is None
    # a gain folded into the templates (what the reference does to the space pair) is not a unit rotation any more
    assert T.tone_model(2.0 * ui, 2.0 * uq)[2] > 0.5


def test_afsk_modem_picks_sliding_sums_only_for_tones():
    import numpy as np
    from pymodem_amd.modems import AFSKModem
    m = AFSKModem(sample_rate=48000, config="1200")
    ui, uq = m.unit_space_correlators()
    t = m._tones(ui, uq)
    assert t is not None and t.tap_dev < 1e-13 and m._tones(ui, uq) is t          # remembered
    m.mark_correlator_i = m.mark_correlator_i * np.hanning(len(m.mark_correlator_i))   # somebody's windowed template
    assert m._tones(ui, uq) is None
