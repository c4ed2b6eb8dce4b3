"""End-to-end parity at BASELINE size under the driver's `-m gpu` run: whole chains of the bench workloads over the full 10-minute
(28.8 M-sample) packet-bearing bench buffer, group executor on the GPU against the oracle in canonical FIR order -- slicer bytes,
stream addresses and packets (payload, address, corrected bytes) must be identical.  fsk_9600 = BASELINE configs[2] (three chains
on one front end), bpsk_300 = configs[1], the eight chains a GPU carries of the qpsk_2400 sweep = configs[4], three of the AFSK gain sweep =
configs[3] (all eight: tests/test_gpu_pipe_bitstream.py).  The oracle side runs on a thread pool beside the GPU (conftest.oracle_chains)."""
import numpy as np
import pytest

import bench
from conftest import oracle_chains
from oracle import oracle as O

pytestmark = pytest.mark.gpu
N = 28_800_000


def _buffer(workload):
    class A:
        pass
    a = A()
    a.samples, a.rate, a.workload, a.buffer = N, 48000, workload, "signal"
    return bench.make_buffer(a)


@pytest.mark.parametrize("workload,chain_ids", [("fsk_9600", [0, 1, 2]), ("bpsk_300", [0]), ("qpsk_2400", [0, 1, 2, 3, 4, 5, 6, 7]),
                                                ("afsk_1200_super_opt", [0, 3, 7])])
def test_chains_at_full_size_match_the_oracle(workload, chain_ids):
    from pymodem_amd import chain_builder as cb, chain_execute as ce
    audio = _buffer(workload)
    factory = bench.WORKLOADS[workload][0]
    lines = [factory(c) for c in chain_ids]
    chains = [cb.build_chain(48000, line) for line in lines]
    wanted = oracle_chains([(line, audio) for line in lines])
    stages = {}
    packets = ce.process_chains_device(chains, audio, stages=stages)
    total = 0
    for k, line in enumerate(lines):
        want = wanted[k].result()
        got = stages["sliced"][k]
        assert len(got.data) > 10000
        assert np.array_equal(got.data, want["slice_data"]), f"{workload} chain {chain_ids[k]}: slicer bytes differ"
        assert np.array_equal(got.address, want["slice_addr"]), f"{workload} chain {chain_ids[k]}: stream addresses differ"
        a = [(p.streamaddress, bytes(bytearray(p.data)), p.BytesCorrected) for p in packets[k]]
        b = [(p.streamaddress, bytes(bytearray(p.data)), p.BytesCorrected) for p in want["packets"]]
        assert a == b, f"{workload} chain {chain_ids[k]}: packets differ"
        total += len(b)
    assert total > 0            # (the inverted fsk_9600 chain decodes nothing, in the reference too)


@pytest.mark.parametrize("workload,check_ids", [("bpsk_300", [0]), ("qpsk_2400", [0, 1, 2, 3, 4, 5, 6, 7])])
def test_batch_engine_at_full_size(workload, check_ids):
    """The carrier-loop batch engine (pymodem_amd.loop_batch: every recording x chain of a run in flight, 110 time chunks) at
    BASELINE size: two different recordings x all the workload's chains per GPU.  Recording 0 = the bench buffer: the checked chains
    against the oracle (bytes, addresses, packets), every chain against the per-recording group executor; recording 1 = the same
    buffer rotated by 12 345 samples (another AGC normalisation, other loop trajectories): every chain against the group executor."""
    from pymodem_amd import chain_builder as cb, chain_execute as ce
    from pymodem_amd.loop_batch import close_engines, process_recordings_device
    audio = _buffer(workload)
    recs = [audio, np.roll(audio, 12345)]
    factory, cpg, _ = bench.WORKLOADS[workload]
    lines = [factory(c) for c in range(cpg)]
    sets = [[cb.build_chain(48000, line) for line in lines] for _ in recs]
    wanted = oracle_chains([(lines[c], audio) for c in check_ids])
    stages = {}
    try:
        packets = process_recordings_device(sets, recs, stages=stages)
    finally:
        close_engines()
    assert stages.get("fused_slicers")          # the slicers ran inside the engine (pm_lbatch_run_sliced): that is the path checked here
    key = lambda pk: [(p.streamaddress, bytes(bytearray(p.data)), p.BytesCorrected) for p in pk]
    for r, rec in enumerate(recs):
        ref_stages = {}
        ref = ce.process_chains_device([cb.build_chain(48000, line) for line in lines], rec, stages=ref_stages)
        for c in range(cpg):
            got, want = stages["sliced"][r][c], ref_stages["sliced"][c]
            assert len(got.data) > 10000
            assert np.array_equal(got.data, want.data) and np.array_equal(got.address, want.address), (workload, r, c)
            assert key(packets[r][c]) == key(ref[c]), (workload, r, c)
    total = 0
    for k, c in enumerate(check_ids):
        want = wanted[k].result()
        got = stages["sliced"][0][c]
        assert np.array_equal(got.data, want["slice_data"]) and np.array_equal(got.address, want["slice_addr"]), (workload, c)
        assert key(packets[0][c]) == key(want["packets"]), (workload, c)
        total += len(want["packets"])
    assert total > 0
