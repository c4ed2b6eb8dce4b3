"""The carrier-loop engine in the shape bench.py times, straight against the oracle (psk.py:162-195, psk.py:705-773, slicer.py:59-107,
slicer.py:193-242).  No tune switch is touched: the run is large enough that the library's own defaults pick
  * the direct loop kernel (`loop_direct_kernel`: every lane a loop, more than 2048 loops in a launch; for BPSK with the AGC in the loop's lane),
  * compute units of their own for the loops and CU-masked front / tail streams (64 loop waves or more), and
  * the row slicers inside the run, on the loops' units (`pm_lbatch_run_sliced`),
which the other engine tests reach only with forced switches at 30 000 samples or through HIP-against-HIP comparisons.  Checked against
`O.run_chain(..., canon=True)`: slicer bytes, stream addresses and packets of (recording, chain) pairs spread over the run -- the first
and last loop and both sides of loop-wave boundaries (lanes 63 | 64 of the loop launch, the middle of the run, the last wave), every
checked recording with audio of its own.  And: configs[4] with all 64 chains of the sweep on one GPU (`bench.wl_qpsk_2400(0..63)`),
chains 8, 31, 62, 63 against the oracle -- carriers 8..63 of the sweep had never been compared with anything."""
import numpy as np
import pytest

import bench
from conftest import noise_i16, oracle_chains
from oracle import oracle as O

pytestmark = pytest.mark.gpu
N = 200_000


def _key(packets):
    return [(int(p.streamaddress), bytes(bytearray(p.data)), int(p.BytesCorrected)) for p in packets]


def _distinct(mode, k, n=N):
    """Recording number k of the checked ones: a generated packet (so that there is something to decode) under noise of its own."""
    from pymodem_amd import siggen
    sig = siggen.recording(mode, 48000, packets=6 if mode.startswith("qpsk") else 2, seed=900 + k, noise_sigma=1500.0 + 200.0 * k, payload_len=(20, 40))[0]
    out = noise_i16(n, seed=7000 + k, sigma=1800.0)
    m = min(n, len(sig))
    out[:m] = np.clip(out[:m].astype(np.int32) + sig[:m], -32768, 32767).astype(np.int16)
    return out


def _run(workload, mode, recordings, nchains, pairs, n=N):
    """One engine run of `recordings` recordings x the workload's first `nchains` chains; the recordings named in `pairs` get audio of
    their own, the others share one noise buffer.  -> checks, total packets over the checked pairs"""
    import pymodem_amd
    from pymodem_amd import chain_builder as cb, loop_batch as lb
    factory = bench.WORKLOADS[workload][0]
    lines = [factory(c) for c in range(nchains)]
    ctx = pymodem_amd.Context.default()
    checked = sorted({r for r, _ in pairs})
    host = {r: _distinct(mode, k, n) for k, r in enumerate(checked)}
    filler = ctx.upload(noise_i16(n, seed=4242, sigma=3000.0))
    dev = {r: ctx.upload(a) for r, a in host.items()}
    ctx.sync()
    audios = [dev.get(r, filler) for r in range(recordings)]
    sets = [[cb.build_chain(48000, line) for line in lines] for _ in range(recordings)]
    wanted = oracle_chains([(lines[c], host[r]) for r, c in pairs])
    stages = {}
    try:
        packets = lb.process_recordings_device(sets, audios, ctx, chunk=65536, stages=stages)
        eng = lb.engine_for([ch[1] for ch in sets[0]], recordings, ctx, 65536)
        loops_have_their_own_units = eng.loop is not None
    finally:
        lb.close_engines()
    # the defaults took the path the bench times
    assert stages.get("fused_slicers"), "the slicers did not run inside the engine"
    assert loops_have_their_own_units, "the loops did not get compute units of their own (CU-masked streams)"
    assert recordings * nchains > 2048                        # pm_loops.hip loop_shape(): the direct kernel from 2049 loops
    total = 0
    for k, (r, c) in enumerate(pairs):
        want = wanted[k].result()
        got = stages["sliced"][r][c]
        assert len(got.data) > 30
        assert np.array_equal(got.data, want["slice_data"]), (workload, r, c, "slicer bytes")
        assert np.array_equal(got.address, want["slice_addr"]), (workload, r, c, "stream addresses")
        assert _key(packets[r][c]) == _key(want["packets"]), (workload, r, c, "packets")
        total += len(want["packets"])
    return total


def test_bpsk_300_engine_in_the_bench_s_shape_equals_the_oracle():
    """4224 recordings x 1 chain: 66 loop waves on units of their own (the Costas loop with the AGC in its lane), direct kernel, row slicers."""
    recordings = 4224
    pairs = [(r, 0) for r in (0, 63, 64, 2111, 2112, 4159, 4160, recordings - 1)]
    total = _run("bpsk_300", "bpsk300_il2p", recordings, 1, pairs)
    assert total >= 4


def test_qpsk_2400_engine_in_the_bench_s_shape_equals_the_oracle():
    """528 recordings x 8 chains = 4224 two-output loops: loop l = recording * 8 + chain, wave boundaries at l = 64 k."""
    recordings = 528
    pairs = [(0, 0), (7, 7), (8, 0), (263, 7), (264, 0), (519, 7), (520, 0), (recordings - 1, 7), (300, 3)]
    total = _run("qpsk_2400", "qpsk2400_il2p", recordings, 8, pairs)
    assert total >= 6


def test_all_64_chains_of_the_qpsk_sweep_on_one_gpu_equal_the_oracle():
    """BASELINE configs[4] whole on one GPU (`also.qpsk_2400.all_64_chains_on_one_gpu`): 72 recordings x 64 chains = 4608 loops; chains
    8, 31, 62, 63 (carriers 1512.5, 1450, 1403.125, 1600 Hz: far from the generated signal's 1500, so mostly the loops' behaviour on
    an off-tune carrier and on noise) and 0, 5 beside them, on three recordings."""
    recordings = 72
    pairs = [(0, 8), (0, 31), (0, 62), (0, 63), (35, 0), (35, 5), (35, 62), (recordings - 1, 63), (recordings - 1, 31)]
    total = _run("qpsk_2400", "qpsk2400_il2p", recordings, 64, pairs, n=120_000)
    assert total >= 1
