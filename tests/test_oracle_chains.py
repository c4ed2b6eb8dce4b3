"""Oracle demod_chain (modem -> slicer -> LFSR -> codec -> de-dup) vs the reference run on the same
inputs: every working bundled config on seeded noise (synth_chains.npz) and the one bundled recording
(wav_chains.npz).  Slicer bytes, addresses, LFSR bytes and packets must be IDENTICAL; FIR-bearing
intermediates within 1e-9 of max|y| (numpy.convolve's summation order is unspecified)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, noise_i16, read_wav_pcm16
from oracle import oracle as O

TOL = 1e-9
MANIFEST = json.load(open(os.path.join(GOLDEN, "synth_chains_manifest.json")))["configs"]
CASES = [(48000, 24000, "48k_s"), (48000, 240000, "48k_l"), (8000, 16000, "8k_s"), (44100, 24000, "44k_s")]


def pk(pkts):
    return (np.array([p.streamaddress for p in pkts], dtype=np.int64), np.array([len(p.data) for p in pkts], dtype=np.int64),
            np.array([p.BytesCorrected for p in pkts], dtype=np.int64), np.array([b for p in pkts for b in p.data], dtype=np.uint8))


def check_chain(g, prefix, r, decim=1):
    d = r["demod"]
    parts = list(zip(d, ("_demod_i", "_demod_q"))) if isinstance(d, tuple) else [(d, "_demod")]
    assert len(parts[0][0]) == int(g[prefix + "_n_demod"])
    for a, k in parts:
        if prefix + k in g.files:
            ref = g[prefix + k]
            assert np.abs(a[::decim] - ref).max() <= TOL * np.abs(ref).max()
    assert np.array_equal(r["slice_data"], g[prefix + "_slice_data"])
    assert np.array_equal(r["slice_addr"], g[prefix + "_slice_addr"])
    assert np.array_equal(r["lfsr"], g[prefix + "_lfsr_data"])
    a, l, c, dd = pk(r["packets"])
    assert np.array_equal(a, g[prefix + "_pkt_addr"]) and np.array_equal(l, g[prefix + "_pkt_len"])
    assert np.array_equal(c, g[prefix + "_pkt_corrected"]) and np.array_equal(dd, g[prefix + "_pkt_data"])


@pytest.mark.parametrize("cfg", sorted(MANIFEST))
def test_synthetic_chains(golden, config_lines, cfg):
    g = golden("synth_chains")
    ran = 0
    for ci, line in enumerate(config_lines(cfg)):
        for rate, n, tag in CASES:
            prefix = f"{cfg[:-5]}__c{ci}__{tag}"
            if prefix + "_n_demod" not in g.files:
                continue
            for canon in ((True,) if tag == "48k_l" else (False, True)):
                check_chain(g, prefix, O.run_chain(O.build_chain(rate, line), noise_i16(n), canon=canon))
                ran += 1
    assert ran >= 3


@pytest.mark.parametrize("cfg", ["afsk_300.json", "afsk_300_pll.json", "afsk_300_ax25.json"])
def test_bundled_recording(golden, config_lines, cfg):
    """SURVEY 8c known answers: 49 good / 6 bad, 48 / 0, 0 / 30."""
    g = golden("wav_chains")
    summ = json.load(open(os.path.join(GOLDEN, "wav_chains_summary.json")))
    rate, audio = read_wav_pcm16(os.path.join(GOLDEN, "afsk_300_il2pc_noise.wav"))
    assert rate == summ["rate"] and len(audio) == summ["n"]
    k = cfg[:-5]
    allp = []
    for ci, line in enumerate(config_lines(cfg)):
        r = O.run_chain(O.build_chain(rate, line), audio, canon=(ci % 2 == 0))
        check_chain(g, f"{k}__c{ci}", r, decim=499)
        for p in r["packets"]:
            p.check()
        allp.append(r["packets"])
    uniq = O.correlate(allp, rate / 40)       # pymodem.py:175
    flat = [p for pl in allp for p in pl]
    assert sum(1 for u in uniq if u.ValidCRC and u.ValidHeader) == summ[k]["good"]
    assert sum(1 for p in flat if not (p.ValidCRC and p.ValidHeader)) == summ[k]["bad"]
    assert np.array_equal(np.array([u.streamaddress for u in uniq], dtype=np.int64), g[k + "__uniq_addr"])
    assert np.array_equal(np.array([u.CalculatedCRC for u in uniq], dtype=np.int64), g[k + "__uniq_crc"])
    assert [list(u.CorrelatedDecoders) for u in uniq] == summ[k]["uniq_decoders"]
    raw = np.array([[int(p.ValidCRC), int(p.ValidHeader), p.CalculatedCRC, p.CarriedCRC] for p in flat], dtype=np.int64).reshape(-1, 4)
    assert np.array_equal(raw, g[k + "__raw_valid"])
