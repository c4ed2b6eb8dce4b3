"""Packet-bearing recordings (tests/golden/signal_chains.npz: audio from pymodem_amd.siggen, every stage output as the
REFERENCE decoded it) through the oracle (CPU) and through the GPU chains; plus encode -> decode round trips of the
generator itself.  AFSK-1200 AX.25/IL2P, GFSK-9600 AX.25 (G3RUH) / IL2P, BPSK-300/1200, QPSK-600/2400/3600 at 48 kHz and an
8-chain AFSK config at 44.1 kHz; noise levels chosen so that RS corrections (6-21 bytes) and a CRC failure occur."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle as O

SUMMARY = json.load(open(os.path.join(GOLDEN, "signal_chains_summary.json")))
CASES = [tuple(c) for c in SUMMARY["cases"]]
IDS = [f"{c[0]}-{c[1][:-5]}-{c[2]}" for c in CASES]


def pk(pkts):
    return (np.array([p.streamaddress for p in pkts], dtype=np.int64), np.array([len(p.data) for p in pkts], dtype=np.int64),
            np.array([p.BytesCorrected for p in pkts], dtype=np.int64), np.array([b for p in pkts for b in p.data], dtype=np.uint8))


def check_stage_outputs(g, prefix, sliced_data, sliced_addr, lfsr, pkts):
    assert np.array_equal(sliced_data, g[prefix + "_slice_data"]) and np.array_equal(sliced_addr, g[prefix + "_slice_addr"]), prefix
    assert np.array_equal(lfsr, g[prefix + "_lfsr_data"]), prefix
    a, l, c, dd = pk(pkts)
    assert np.array_equal(a, g[prefix + "_pkt_addr"]) and np.array_equal(l, g[prefix + "_pkt_len"]), prefix
    assert np.array_equal(c, g[prefix + "_pkt_corrected"]) and np.array_equal(dd, g[prefix + "_pkt_data"]), prefix


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_oracle_decodes_like_the_reference(golden, config_lines, case):
    mode, cfg, rate = case[0], case[1], case[2]
    g = golden("signal_chains")
    name = f"{mode}__{cfg[:-5]}__{rate}"
    audio = g[name + "__audio"]
    allp = []
    for ci, line in enumerate(config_lines(cfg)):
        r = O.run_chain(O.build_chain(rate, line), audio, canon=True)
        prefix = f"{name}__c{ci}"
        d = r["demod"]
        parts = list(zip(d, ("_demod_i", "_demod_q"))) if isinstance(d, tuple) else [(d, "_demod")]
        for arr, k in parts:
            ref = g[prefix + k]
            assert np.abs(arr[::97] - ref).max() <= 1e-9 * np.abs(ref).max()
        check_stage_outputs(g, prefix, r["slice_data"], r["slice_addr"], r["lfsr"], r["packets"])
        for p in r["packets"]:
            p.check()
        allp.append(r["packets"])
    uniq = O.correlate(allp, rate / 40)
    res = SUMMARY["results"][name]
    assert sum(1 for u in uniq if u.ValidCRC and u.ValidHeader) == res["good"]
    assert np.array_equal(np.array([u.streamaddress for u in uniq], dtype=np.int64), g[name + "__uniq_addr"])
    assert [list(u.CorrelatedDecoders) for u in uniq] == res["uniq_decoders"]


def test_goldens_exercise_rs_corrections_and_crc_failures():
    res = SUMMARY["results"]
    assert sum(c["corrected"] for r in res.values() for c in r["chains"]) >= 40
    assert any(r["bad"] > 0 for r in res.values())
    assert all(r["good"] >= 2 for r in res.values())


# ---- generator round trips (host only) -------------------------------------------------------------------------------------
def test_lfsr_scramble_is_the_inverse_of_the_stream_stage():
    from pymodem_amd import siggen
    from pymodem_amd.data_classes import AddressedArray
    from pymodem_amd.lfsr import LFSR
    rng = np.random.default_rng(0)
    for poly, inv in [(0x1, False), (0x1, True), (0x3, True), (0x63003, True), (0x211, False)]:
        want = rng.integers(0, 2, 8 * 500).tolist()
        sent = siggen.lfsr_scramble(want, poly, inv)
        data = np.packbits(np.array(sent, dtype=np.uint8))
        got = LFSR(poly=poly, invert=inv).stream_unscramble_8bit(AddressedArray(data, np.arange(len(data))))
        assert np.unpackbits(got.data).tolist() == want


def test_il2p_and_ax25_framers_round_trip_through_the_native_codecs():
    from pymodem_amd import siggen
    from pymodem_amd.codecs import AX25Codec, IL2PCodec
    from pymodem_amd.data_classes import AddressedArray
    rng = np.random.default_rng(1)
    for n in [0, 1, 17, 100, 238, 239, 240, 478, 500, 1023]:
        info = rng.integers(0, 256, n).tolist()
        frame = siggen.ax25_ui_frame("DEST", "SRC", info, dest_ssid=3, src_ssid=9)
        for crc in (True, False):
            bits = [0, 1] * 20 + siggen.il2p_frame_bits("DEST", "SRC", info, dest_ssid=3, src_ssid=9, trailing_crc=crc) + [0] * 24
            bits += [0] * (-len(bits) % 8)
            data = np.packbits(np.array(bits, dtype=np.uint8))
            # flip a few bits inside the payload area: RS must repair them
            if n >= 100:
                for pos in rng.choice(np.arange(40, len(data) - 30), 3, replace=False):
                    data[pos] ^= 0x10
            pk_ = IL2PCodec(ident="x", crc=crc).decode(AddressedArray(data, np.arange(len(data))))
            assert len(pk_) == 1 and pk_[0].data[:-2] == frame, (n, crc)
            assert pk_[0].CalcCRC()
            assert pk_[0].BytesCorrected == (3 if n >= 100 else 0)
        if 1 <= n <= 1000:          # the reference drops AX.25 frames longer than 1023 bytes (ax25.py:15,44-50)
            bits = siggen.ax25_hdlc_bits(frame)
            bits += [0] * (-len(bits) % 8)
            data = np.packbits(np.array(bits, dtype=np.uint8))
            pk_ = AX25Codec(ident="x").decode(AddressedArray(data, np.arange(len(data))))
            assert len(pk_) == 1 and pk_[0].data[:-2] == frame and pk_[0].CalcCRC()


@pytest.mark.parametrize("mode,cfg,ci,rate", [("afsk1200_ax25", "afsk_1200.json", 0, 48000), ("afsk1200_il2p", "afsk_1200.json", 2, 22050),
                                              ("fsk9600_ax25", "fsk_9600.json", 2, 48000), ("fsk9600_il2p", "fsk_9600.json", 0, 96000),
                                              ("bpsk1200_il2p", "bpsk_1200.json", 0, 44100), ("qpsk2400_il2p", "qpsk_2400.json", 1, 48000)])
def test_generated_signals_decode_in_the_oracle(config_lines, mode, cfg, ci, rate):
    """encode -> modulate -> (oracle) demodulate -> decode returns exactly the frames that were sent."""
    from pymodem_amd import siggen
    audio, frames = siggen.recording(mode, rate, packets=3, seed=5, noise_sigma=300.0, payload_len=(10, 40))
    r = O.run_chain(O.build_chain(rate, config_lines(cfg)[ci]), audio)
    got = []
    for p in r["packets"]:
        p.check()
        if p.ValidCRC:
            got.append([int(b) for b in p.data[:-2]])
    assert got == frames


# ---- GPU ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_gpu_chains_decode_like_the_reference(golden, config_lines, case):
    from pymodem_amd import chain_builder as cb, chain_execute as ce, dist as pdist
    mode, cfg, rate = case[0], case[1], case[2]
    g = golden("signal_chains")
    name = f"{mode}__{cfg[:-5]}__{rate}"
    audio = g[name + "__audio"]
    lines = config_lines(cfg)
    # (1) stage by stage through the reference-shaped API
    for ci, line in enumerate(lines):
        chain = cb.build_chain(rate, line)
        demod = chain[1].demod(audio)
        sliced = chain[2].slice(demod)
        lf = chain[3].stream_unscramble_8bit(sliced)
        pkts = chain[4].decode(lf)
        check_stage_outputs(g, f"{name}__c{ci}", sliced.data, sliced.address, lf.data, pkts)
        want = O.run_chain(O.build_chain(rate, line), audio, canon=True)["demod"]
        got = [demod.i_data, demod.q_data] if hasattr(demod, "i_data") else [demod]
        for a, b in zip(got, list(want) if isinstance(want, tuple) else [want]):
            assert np.array_equal(a, b)                       # bit-exact against the oracle
    # (2) the group executor + de-dup
    chains = [cb.build_chain(rate, line) for line in lines]
    stages = {}
    pkts = ce.process_chains_device(chains, audio, stages)
    for ci in range(len(lines)):
        a, l, c, dd = pk(pkts[ci])
        prefix = f"{name}__c{ci}"
        assert np.array_equal(stages["sliced"][ci].data, g[prefix + "_slice_data"])
        assert np.array_equal(a, g[prefix + "_pkt_addr"]) and np.array_equal(dd, g[prefix + "_pkt_data"]) and np.array_equal(c, g[prefix + "_pkt_corrected"])
    arr = pdist.correlate(dict(enumerate(pkts)), len(lines), rate / 40)
    res = SUMMARY["results"][name]
    assert arr.CountGood() == res["good"] and arr.CountBad() == res["bad"]
    assert np.array_equal(np.array([p.streamaddress for p in arr.unique_packet_array], dtype=np.int64), g[name + "__uniq_addr"])
    assert [list(p.CorrelatedDecoders) for p in arr.unique_packet_array] == res["uniq_decoders"]
