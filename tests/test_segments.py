"""Stage objects are stateful, as in the reference: a second process_chain on the same chain continues the first (AGC envelope,
carrier loop, slicer clock / open byte / address count, LFSR register, codec state machine); only the FIRs start afresh, as
numpy.convolve('valid') does there.  Goldens: the reference fed a generated recording in two pieces
(tests/golden/make_goldens.py segments)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle as O

CASES = ["afsk_1200", "bpsk_300", "fsk_9600", "qpsk_2400"]


def check_segment(g, prefix, slice_data, slice_addr, pkts):
    assert np.array_equal(slice_data, g[prefix + "_slice_data"]) and np.array_equal(slice_addr, g[prefix + "_slice_addr"])
    assert len(pkts) == int(g[prefix + "_pkt_n"])
    assert np.array_equal(np.array([p.streamaddress for p in pkts], dtype=np.int64), g[prefix + "_pkt_addr"])
    assert np.array_equal(np.array([b for p in pkts for b in p.data], dtype=np.uint8), g[prefix + "_pkt_data"])


@pytest.mark.parametrize("tag", CASES)
def test_oracle_continues_like_the_reference(golden, config_lines, tag):
    g = golden("segments")
    cut = json.load(open(os.path.join(GOLDEN, "segments_summary.json")))[tag]["cut"]
    audio = g[tag + "__audio"]
    chain = O.build_chain(48000, config_lines(tag + ".json")[0])
    for k, seg in enumerate((audio[:cut], audio[cut:])):
        r = O.run_chain(chain, seg)
        check_segment(g, f"{tag}__seg{k}", r["slice_data"], r["slice_addr"], r["packets"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_gpu_chain_continues_like_the_reference(golden, config_lines, tag):
    from pymodem_amd import chain_builder as cb, chain_execute as ce
    g = golden("segments")
    cut = json.load(open(os.path.join(GOLDEN, "segments_summary.json")))[tag]["cut"]
    audio = g[tag + "__audio"]
    line = config_lines(tag + ".json")[0]
    for run in ("stages", "device", "native"):
        chain = cb.build_chain(48000, line)
        nc = ce.NativeChain(chain[1], chain[2]) if run == "native" else None
        for k, seg in enumerate((audio[:cut], audio[cut:])):
            if run == "stages":
                sliced = chain[2].slice(chain[1].demod(seg))
            elif run == "device":
                sliced = chain[2].slice(chain[1].demod_signs(seg))
            else:
                sliced = nc.run(seg)                      # the C chain object carries AGC, loop and slicer state itself
            pkts = chain[4].decode(chain[3].stream_unscramble_8bit(sliced))
            check_segment(g, f"{tag}__seg{k}", sliced.data, sliced.address, pkts)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,ci", [("afsk_1200", 0), ("afsk_1200", 2), ("afsk_1200_ax25_super_opt", 5), ("fsk_9600", 0), ("fsk_9600", 2), ("fsk_4800", 0)])
@pytest.mark.parametrize("run", ["stages", "signs", "group", "native"])
def test_carried_fir_history_makes_pieces_equal_the_whole(golden, config_lines, cfg, ci, run):
    """SURVEY 8f-3, opt-in carry_history: the 240 000-sample golden input fed in three uneven pieces gives the reference's SINGLE-CALL
    slicer bytes, stream addresses, LFSR bytes and packets (tests/golden/synth_chains.npz), for AFSK and FSK chains, through the
    stage objects, the sign-bitmap path, the group executor and the whole-chain C entry point.  (The first piece is shorter than
    some filters' history, the second starts mid-word of the bitmap.)"""
    from conftest import noise_i16
    from pymodem_amd import chain_builder as cb, chain_execute as ce
    from pymodem_amd.data_classes import AddressedArray
    g = golden("synth_chains")
    prefix = f"{cfg}__c{ci}__48k_l"
    if prefix + "_slice_data" not in g.files:
        pytest.skip("no golden for this chain")
    line = config_lines(cfg + ".json")[ci]
    audio = noise_i16(240000)
    cuts = [0, 173, 100003, 240000]
    chain = cb.build_chain(48000, line)
    chain[1].carry_history = True
    nc = ce.NativeChain(chain[1], chain[2]) if run == "native" else None
    data, addr, pkts, lfsr = [], [], [], []
    for a, b in zip(cuts, cuts[1:]):
        seg = audio[a:b]
        if run == "stages":
            sliced = chain[2].slice(chain[1].demod(seg))
        elif run == "signs":
            sliced = chain[2].slice(chain[1].demod_signs(seg))
        elif run == "group":
            st = {}
            ce.process_chains_device([chain], seg, stages=st)       # (also runs stream and codec: they are fed again below from a copy)
            sliced = st["sliced"][0]
            data.append(np.array(sliced.data)); addr.append(np.array(sliced.address))
            continue
        else:
            sliced = nc.run(seg)
        data.append(np.array(sliced.data)); addr.append(np.array(sliced.address))
        lf = chain[3].stream_unscramble_8bit(sliced)
        lfsr.append(np.array(lf.data))
        pkts += chain[4].decode(lf)
    data, addr = np.concatenate(data), np.concatenate(addr)
    assert np.array_equal(data, g[prefix + "_slice_data"]) and np.array_equal(addr, g[prefix + "_slice_addr"])
    if run == "group":
        return
    assert np.array_equal(np.concatenate(lfsr), g[prefix + "_lfsr_data"])
    assert np.array_equal(np.array([p.streamaddress for p in pkts], dtype=np.int64), g[prefix + "_pkt_addr"])
    assert np.array_equal(np.array([b for p in pkts for b in p.data], dtype=np.uint8), g[prefix + "_pkt_data"])
    # without the flag the same pieces are the reference's per-call behaviour: fewer samples reach the slicer
    plain = cb.build_chain(48000, line)
    n = sum(len(plain[2].slice(plain[1].demod(audio[a:b]))) for a, b in zip(cuts[1:], cuts[2:]))
    assert n <= len(data)


PSK_HISTORY = ["bpsk_300", "qpsk_2400", "afsk_300_pll"]


def _history_pieces(golden, tag):
    g = golden("psk_history")
    summ = json.load(open(os.path.join(GOLDEN, "psk_history_summary.json")))[tag]
    audio = g[tag + "__audio"]
    return g, summ["rate"], [audio[a:b] for a, b in zip(summ["cuts"][:-1], summ["cuts"][1:])]


@pytest.mark.parametrize("tag", PSK_HISTORY)
def test_oracle_carries_fir_history_through_the_carrier_loop_modems(golden, config_lines, tag):
    """carry_history for BPSK / MPSK / AFSK-PLL: every FIR of the cascade continues from the last M - 1 samples of its input, AGC
    envelope and loop registers live on, AGC.apply normalises by each call's own maximum (agc.py:67) -- against what the reference's
    primitives give when fed that way (tests/golden/make_goldens.py psk_history: three uneven pieces, the first shorter than the
    filters)."""
    g, rate, pieces = _history_pieces(golden, tag)
    chain = O.build_chain(rate, config_lines(tag + ".json")[0])
    chain[0].carry_history = True                  # (the oracle's chain is (modem, slicer, stream, codec))
    total = 0
    for k, seg in enumerate(pieces):
        r = O.run_chain(chain, seg)
        check_segment(g, f"{tag}__seg{k}", r["slice_data"], r["slice_addr"], r["packets"])
        total += len(r["slice_data"])
    assert total > 100


@pytest.mark.gpu
@pytest.mark.parametrize("tag", PSK_HISTORY)
@pytest.mark.parametrize("run", ["stages", "signs", "group", "native"])
def test_gpu_carries_fir_history_through_the_carrier_loop_modems(golden, config_lines, tag, run):
    from pymodem_amd import chain_builder as cb, chain_execute as ce
    g, rate, pieces = _history_pieces(golden, tag)
    chain = cb.build_chain(rate, config_lines(tag + ".json")[0])
    chain[1].carry_history = True
    oracle = O.build_chain(rate, config_lines(tag + ".json")[0])
    oracle[0].carry_history = True
    nc = ce.NativeChain(chain[1], chain[2]) if run == "native" else None      # pm_chain with PM_CHAIN_CARRY_HISTORY: tails on the device
    for k, seg in enumerate(pieces):
        if run == "native":
            sliced = nc.run(seg)
            pkts = chain[4].decode(chain[3].stream_unscramble_8bit(sliced))
        elif run == "stages":
            sliced = chain[2].slice(chain[1].demod(seg))
            pkts = chain[4].decode(chain[3].stream_unscramble_8bit(sliced))
        elif run == "signs":
            sliced = chain[2].slice(chain[1].demod_signs(seg))
            pkts = chain[4].decode(chain[3].stream_unscramble_8bit(sliced))
        else:
            st = {}
            pkts = ce.process_chains_device([chain], seg, stages=st)[0]
            sliced = st["sliced"][0]
        check_segment(g, f"{tag}__seg{k}", sliced.data, sliced.address, pkts)
        want = O.run_chain(oracle, seg, canon=True)                     # and bit for bit against the oracle in the kernels' FIR order
        assert np.array_equal(sliced.data, want["slice_data"]) and np.array_equal(sliced.address, want["slice_addr"])
