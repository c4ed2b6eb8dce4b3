"""QPSKModem (psk.py:197-476, chain_builder type 'qpsk'): oracle pinned to goldens made by the reference
(tests/golden/make_goldens.py qpsk), GPU path against the oracle and the goldens."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, noise_i16
from oracle import oracle as O

CASES = [("600", "qpsk_600", 8000), ("600", "qpsk_600", 48000), ("2400", "qpsk_2400", 48000), ("3600", "qpsk_3600", 48000)]


def line(preset, slicer):
    return {"object_name": f"QPSK {preset}", "object_type": "demod_chain", "modem": {"type": "qpsk", "config": preset, "options": {}},
            "slicer": {"type": "quadrature", "config": slicer, "options": {}},
            "stream": {"type": "lfsr", "options": {"poly": "0x1", "invert": "False"}},
            "codec": {"type": "il2p", "options": {"crc": "yes", "disable_rs": "no", "min_dist": "0", "sync_tol": "0"}}}


def packets_equal(pkts, g, prefix):
    assert len(pkts) == int(g[prefix + "_pkt_n"])
    assert np.array_equal(np.array([p.streamaddress for p in pkts], dtype=np.int64), g[prefix + "_pkt_addr"])
    assert np.array_equal(np.array([b for p in pkts for b in p.data], dtype=np.uint8), g[prefix + "_pkt_data"])
    assert np.array_equal(np.array([p.BytesCorrected for p in pkts], dtype=np.int64), g[prefix + "_pkt_corrected"])


@pytest.mark.parametrize("preset,slicer,rate", CASES)
def test_oracle_matches_the_reference(golden, preset, slicer, rate):
    g = golden("qpsk_modem")
    tag = f"qpsk{preset}_{rate}"
    m = O.build_chain(rate, line(preset, slicer))
    assert np.array_equal(m[0].input_bpf, g[tag + "__taps_bpf"]) and np.array_equal(m[0].rrc, g[tag + "__taps_rrc"])
    res = O.run_chain(m, noise_i16(24000))
    di, dq = res["demod"]
    scale = max(np.abs(g[tag + "__noise_demod_i"]).max(), np.abs(g[tag + "__noise_demod_q"]).max())
    assert len(di) == int(g[tag + "__noise_n_demod"])
    assert np.abs(di - g[tag + "__noise_demod_i"]).max() <= 1e-9 * scale and np.abs(dq - g[tag + "__noise_demod_q"]).max() <= 1e-9 * scale
    assert np.array_equal(res["slice_data"], g[tag + "__noise_slice_data"]) and np.array_equal(res["slice_addr"], g[tag + "__noise_slice_addr"])
    # the generated recording: every packet the reference decodes, byte for byte
    summary = json.load(open(os.path.join(GOLDEN, "qpsk_modem_summary.json")))
    conj = int(summary[tag]["kept_conj"])
    assert summary[f"{tag}__sig{conj}"]["good_crc"] == 4
    res = O.run_chain(O.build_chain(rate, line(preset, slicer)), g[tag + "__sig_audio"])
    prefix = f"{tag}__sig{conj}"
    assert np.array_equal(res["slice_data"], g[prefix + "_slice_data"]) and np.array_equal(res["slice_addr"], g[prefix + "_slice_addr"])
    packets_equal(res["packets"], g, prefix)


def test_factory_builds_the_qpsk_modem_without_a_gpu():
    from pymodem_amd import chain_builder as cb
    m = cb.ModemConfigurator(48000, {"type": "qpsk", "config": "2400", "options": {"carrier_freq": "1750"}})
    assert type(m).__name__ == "QPSKModem" and m.carrier_freq == 1750.0 and len(m.rrc_taps) == 121
    o = O.QPSKModem(48000, "2400", {"carrier_freq": "1750"})
    assert np.array_equal(m.input_bpf, o.input_bpf) and np.array_equal(m.rrc_taps, o.rrc)
    assert (m._loop.bb0, m._loop.bb1, m._loop.ba1) == tuple(o.branch[:3])
    with pytest.raises(AttributeError):
        cb.ModemConfigurator(48000, {"type": "qpsk", "config": "1200", "options": {}})


@pytest.mark.gpu
@pytest.mark.parametrize("preset,slicer,rate", CASES)
def test_gpu_qpsk_chain(golden, preset, slicer, rate):
    from pymodem_amd import chain_builder as cb, chain_execute as ce
    g = golden("qpsk_modem")
    tag = f"qpsk{preset}_{rate}"
    summary = json.load(open(os.path.join(GOLDEN, "qpsk_modem_summary.json")))
    prefix = f"{tag}__sig{int(summary[tag]['kept_conj'])}"
    for audio, gp in ((noise_i16(24000), tag + "__noise"), (g[tag + "__sig_audio"], prefix)):
        want = O.run_chain(O.build_chain(rate, line(preset, slicer)), audio, canon=True)
        ch = cb.build_chain(rate, line(preset, slicer))
        d = ch[1].demod(audio)
        assert np.array_equal(d.i_data, want["demod"][0]) and np.array_equal(d.q_data, want["demod"][1])       # bit-exact vs the oracle
        for run in (ce.process_chain, ce.process_chain_device):
            ch = cb.build_chain(rate, line(preset, slicer))
            packets_equal(run(ch, audio), g, gp)
        ch = cb.build_chain(rate, line(preset, slicer))
        got = ce.NativeChain(ch[1], ch[2]).run(audio)                                                           # whole-chain C entry
        assert np.array_equal(got.data, g[gp + "_slice_data"]) and np.array_equal(got.address, g[gp + "_slice_addr"])
    ch2 = [cb.build_chain(rate, line(preset, slicer)) for _ in range(2)]
    both = ce.process_chains_device(ch2, g[tag + "__sig_audio"])                                              # group executor
    for pk in both:
        packets_equal(pk, g, prefix)
