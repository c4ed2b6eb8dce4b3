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
