"""GPU parity at chain level: the drop-in stage objects (pymodem_amd.chain_builder / chain_execute) against
(a) the oracle on the same inputs -- demodulated stream bit-exact, bytes/addresses/packets identical -- and
(b) the committed reference goldens -- slicer bytes, addresses, LFSR bytes and packets identical, FIR-bearing
intermediates within 1e-9 of max|y| (numpy.convolve's order is unspecified)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, noise_i16, read_wav_pcm16
from oracle import oracle as O

pytestmark = pytest.mark.gpu
MANIFEST = json.load(open(os.path.join(GOLDEN, "synth_chains_manifest.json")))["configs"]
CASES = [(48000, 24000, "48k_s"), (48000, 240000, "48k_l"), (8000, 16000, "8k_s"), (44100, 24000, "44k_s")]


def gpu_chain(rate, line, audio):
    from pymodem_amd import chain_builder as cb
    chain = cb.build_chain(rate, line)
    demod = chain[1].demod(audio)
    sliced = chain[2].slice(demod)
    lf = chain[3].stream_unscramble_8bit(sliced)
    pkts = chain[4].decode(lf)
    return demod, sliced, lf, pkts


def demod_parts(d):
    if isinstance(d, tuple):
        return list(d)
    if hasattr(d, "i_data"):
        return [np.asarray(d.i_data), np.asarray(d.q_data)]
    return [np.asarray(d)]


def pk(pkts):
    return (np.array([p.streamaddress for p in pkts], dtype=np.int64), np.array([len(p.data) for p in pkts], dtype=np.int64),
            np.array([p.BytesCorrected for p in pkts], dtype=np.int64), np.array([b for p in pkts for b in p.data], dtype=np.uint8))


@pytest.mark.parametrize("cfg", sorted(MANIFEST))
def test_synthetic_chains(golden, config_lines, cfg):
    g = golden("synth_chains")
    for ci, line in enumerate(config_lines(cfg)):
        for rate, n, tag in CASES:
            prefix = f"{cfg[:-5]}__c{ci}__{tag}"
            if prefix + "_n_demod" not in g.files:
                continue
            audio = noise_i16(n)
            demod, sliced, lf, pkts = gpu_chain(rate, line, audio)
            want = O.run_chain(O.build_chain(rate, line), audio, canon=True)
            # (a) against the oracle: everything identical, floats included
            for got, ref in zip(demod_parts(demod), demod_parts(want["demod"])):
                assert np.array_equal(got, ref), (prefix, "demod vs oracle", np.abs(got - ref).max())
            assert np.array_equal(sliced.data, want["slice_data"]) and np.array_equal(sliced.address, want["slice_addr"]), prefix
            assert np.array_equal(lf.data, want["lfsr"]), prefix
            for a, b in zip(pk(pkts), pk(want["packets"])):
                assert np.array_equal(a, b), prefix
            # (b) against the reference's goldens
            keys = ["_demod_i", "_demod_q"] if len(demod_parts(demod)) == 2 else ["_demod"]
            for got, k in zip(demod_parts(demod), keys):
                if prefix + k in g.files:
                    ref = g[prefix + k]
                    assert np.abs(got - ref).max() <= 1e-9 * np.abs(ref).max(), (prefix, k)
            assert np.array_equal(sliced.data, g[prefix + "_slice_data"]) and np.array_equal(sliced.address, g[prefix + "_slice_addr"]), prefix
            assert np.array_equal(lf.data, g[prefix + "_lfsr_data"]), prefix
            a, l, c, dd = pk(pkts)
            assert np.array_equal(a, g[prefix + "_pkt_addr"]) and np.array_equal(l, g[prefix + "_pkt_len"]), prefix
            assert np.array_equal(c, g[prefix + "_pkt_corrected"]) and np.array_equal(dd, g[prefix + "_pkt_data"]), prefix


@pytest.mark.parametrize("cfg", ["afsk_300.json", "afsk_300_pll.json", "afsk_300_ax25.json"])
def test_bundled_recording(golden, config_lines, cfg):
    """The reference's known answers on its one bundled recording: 49 good / 6 bad, 48 / 0, 0 / 30."""
    from pymodem_amd import chain_builder as cb, chain_execute as ce
    from pymodem_amd.packet_meta import PacketMetaArray
    g = golden("wav_chains")
    summ = json.load(open(os.path.join(GOLDEN, "wav_chains_summary.json")))
    rate, audio = read_wav_pcm16(os.path.join(GOLDEN, "afsk_300_il2pc_noise.wav"))
    k = cfg[:-5]
    results = PacketMetaArray()
    for ci, line in enumerate(config_lines(cfg)):
        prefix = f"{k}__c{ci}"
        chain = cb.build_chain(rate, line)
        stages = {}
        pkts = ce.process_chain_device(chain, audio, stages)
        assert np.array_equal(stages["sliced"].data, g[prefix + "_slice_data"]), prefix
        assert np.array_equal(stages["sliced"].address, g[prefix + "_slice_addr"]), prefix
        assert np.array_equal(stages["descrambled"].data, g[prefix + "_lfsr_data"]), prefix
        a, l, c, dd = pk(pkts)
        assert np.array_equal(a, g[prefix + "_pkt_addr"]) and np.array_equal(dd, g[prefix + "_pkt_data"]), prefix
        assert np.array_equal(c, g[prefix + "_pkt_corrected"]), prefix
        # host-returning demod on the same object type: decimated samples against the reference
        d = cb.build_chain(rate, line)[1].demod(audio)
        assert len(d) == int(g[prefix + "_n_demod"])
        ref = g[prefix + "_demod"]
        assert np.abs(d[::499] - ref).max() <= 1e-9 * np.abs(ref).max(), prefix
        results.add(pkts)
    results.CalcCRCs()
    results.Correlate(address_distance=rate / 40)
    assert results.CountGood() == summ[k]["good"] and results.CountBad() == summ[k]["bad"]
    u = results.unique_packet_array
    assert np.array_equal(np.array([p.streamaddress for p in u], dtype=np.int64), g[k + "__uniq_addr"])
    assert np.array_equal(np.array([p.CalculatedCRC for p in u], dtype=np.int64), g[k + "__uniq_crc"])
    assert [list(p.CorrelatedDecoders) for p in u] == summ[k]["uniq_decoders"]


@pytest.mark.parametrize("cfg", ["afsk_1200_ax25_super_opt.json", "qpsk_2400.json", "fsk_9600.json", "bpsk_300.json", "afsk_1200.json"])
def test_group_executor_matches_reference(golden, config_lines, cfg):
    """process_chains_device (shared front ends, batched carrier loops and slicers, threaded host stages) must give
    exactly what the reference gives chain by chain: slicer bytes/addresses and packets of the 240 000-sample goldens."""
    from pymodem_amd import chain_builder as cb, chain_execute as ce
    g = golden("synth_chains")
    lines = config_lines(cfg)
    audio = noise_i16(240000)
    chains = [cb.build_chain(48000, line) for line in lines]
    stages = {}
    pkts = ce.process_chains_device(chains, audio, stages)
    assert len(pkts) == len(lines)
    for ci in range(len(lines)):
        prefix = f"{cfg[:-5]}__c{ci}__48k_l"
        assert np.array_equal(stages["sliced"][ci].data, g[prefix + "_slice_data"]), prefix
        assert np.array_equal(stages["sliced"][ci].address, g[prefix + "_slice_addr"]), prefix
        a, l, c, dd = pk(pkts[ci])
        assert np.array_equal(a, g[prefix + "_pkt_addr"]) and np.array_equal(dd, g[prefix + "_pkt_data"]), prefix
        assert np.array_equal(c, g[prefix + "_pkt_corrected"]), prefix


def test_group_executor_on_bundled_recording(golden, config_lines):
    """afsk_300.json mixes correlator and PLL modems: 49 unique good packets, 6 rejected."""
    from pymodem_amd import chain_builder as cb, chain_execute as ce, dist as pdist
    g = golden("wav_chains")
    rate, audio = read_wav_pcm16(os.path.join(GOLDEN, "afsk_300_il2pc_noise.wav"))
    lines = config_lines("afsk_300.json")
    for _ in range(2):          # twice: work buffers are reused from run to run
        chains = [cb.build_chain(rate, line) for line in lines]
        pkts = ce.process_chains_device(chains, audio)
        for ci in range(len(lines)):
            a, l, c, dd = pk(pkts[ci])
            assert np.array_equal(a, g[f"afsk_300__c{ci}_pkt_addr"]) and np.array_equal(dd, g[f"afsk_300__c{ci}_pkt_data"])
        arr = pdist.correlate(dict(enumerate(pkts)), len(lines), rate / 40)
        assert arr.CountGood() == 49 and arr.CountBad() == 6
        assert np.array_equal(np.array([p.streamaddress for p in arr.unique_packet_array], dtype=np.int64), g["afsk_300__uniq_addr"])


def test_table_path_on_bundled_recording(golden, config_lines):
    """process_chains_table -> PacketTable.correlate: same 49 / 6 and the same unique packets, without PacketMeta objects."""
    from pymodem_amd import chain_builder as cb, chain_execute as ce, dist as pdist
    g = golden("wav_chains")
    rate, audio = read_wav_pcm16(os.path.join(GOLDEN, "afsk_300_il2pc_noise.wav"))
    lines = config_lines("afsk_300.json")
    chains = [cb.build_chain(rate, line) for line in lines]
    rows = ce.process_chains_table(chains, audio)
    table = pdist.gather_rows(rows, len(lines), [l["object_name"] for l in lines]).correlate(rate / 40)
    assert table.CountGood() == 49 and table.CountBad() == 6
    assert np.array_equal(table.rows["streamaddress"][table.unique_idx], g["afsk_300__uniq_addr"])
    for ci in range(len(lines)):
        assert np.array_equal(rows[ci]["streamaddress"], g[f"afsk_300__c{ci}_pkt_addr"])


def test_work_buffers_do_not_accumulate(config_lines):
    """Stage objects are single-use; their pooled work buffers must go with them (stand-alone use) or be reused (group runs)."""
    import gc
    import pymodem_amd
    from pymodem_amd import chain_builder as cb, chain_execute as ce
    ctx = pymodem_amd.Context.default()
    audio = noise_i16(60000)
    lines = config_lines("afsk_1200.json")
    sizes = []
    for _ in range(4):
        chain = cb.build_chain(48000, lines[0])
        ce.process_chain(chain, audio)
        ce.process_chains_device([cb.build_chain(48000, l) for l in lines], audio)
        del chain
        gc.collect()
        sizes.append(len(ctx._pool))
    assert sizes[1] == sizes[2] == sizes[3], sizes


def test_recording_pipeline_matches_one_at_a_time(golden, config_lines):
    """RecordingPipeline (demod | slice on a side stream | host, overlapped across recordings) gives exactly what decoding the
    recordings one at a time gives: different recordings in flight must not touch each other's double-buffered bitmaps."""
    from pymodem_amd import chain_builder as cb, chain_execute as ce, dist as pdist
    g = golden("wav_chains")
    rate, wav = read_wav_pcm16(os.path.join(GOLDEN, "afsk_300_il2pc_noise.wav"))
    lines = config_lines("afsk_300.json")
    names = [l["object_name"] for l in lines]
    recordings = [wav, wav[::-1].copy(), noise_i16(len(wav) // 2), wav, wav[1000:].copy(), wav]
    want = [ce.process_chains_table([cb.build_chain(rate, l) for l in lines], r) for r in recordings]
    pipe = ce.RecordingPipeline()
    futures = [pipe.submit([cb.build_chain(rate, l) for l in lines], r) for r in recordings]
    got = [f.result() for f in futures]
    pipe.close()
    for w, rows in zip(want, got):
        for ci in range(len(lines)):
            assert np.array_equal(w[ci], rows[ci])
    table = pdist.gather_rows(dict(enumerate(got[-1])), len(lines), names).correlate(rate / 40)
    assert table.CountGood() == 49 and table.CountBad() == 6
    for ci in range(len(lines)):
        assert np.array_equal(got[0][ci]["streamaddress"], g[f"afsk_300__c{ci}_pkt_addr"])


def test_recording_pipeline_unordered_tail_drain_and_chain_ids(config_lines):
    """submit(..., unordered=True, chain_ids=...): finish and post run on the host-stage thread and the codecs stamp their packets
    with the chain's place in the config; drain() waits for everything and leaves the executor usable.  Results as one at a time."""
    from pymodem_amd import chain_builder as cb, chain_execute as ce, siggen
    lines = config_lines("afsk_1200_ax25_super_opt.json")
    kinds = [siggen.recording("afsk1200_ax25", 48000, packets=4, seed=s, noise_sigma=500.0, payload_len=(20, 60))[0] for s in (5, 6)]
    want = [ce.process_chains_table([cb.build_chain(48000, l) for l in lines], a) for a in kinds]
    ids = [10 + c for c in range(len(lines))]
    pipe = ce.RecordingPipeline()
    try:
        for rnd in range(2):                                   # two rounds through the same executor with a drain in between
            futures, fin = [], []
            for k in range(9):
                futures.append((k % 2, pipe.submit([cb.build_chain(48000, l) for l in lines], kinds[k % 2], finish=lambda rows: (fin.append(1), rows)[1],
                                                   post=lambda rows: [r.copy() for r in rows], chain_ids=ids, unordered=True)))
            pipe.drain()
            assert len(fin) == 9 and all(f.done() for _, f in futures)
            for which, f in futures:
                rows = f.result()
                for ci in range(len(lines)):
                    r, w = rows[ci], want[which][ci]
                    assert len(r) == len(w) and np.array_equal(r["data"], w["data"]) and np.array_equal(r["streamaddress"], w["streamaddress"])
                    assert (r["source_decoder"] == ids[ci]).all()
            assert sum(len(r) for r in want[0].values()) > 0
    finally:
        pipe.close()


def test_work_buffers_come_from_arenas_and_are_reused():
    """Context.scratch carves its buffers out of large device blocks: distinct tags get distinct ranges, a released owner's range is
    handed out again for the same size, and what is written through one view is read back through the next."""
    import gc
    import pymodem_amd
    ctx = pymodem_amd.Context(0)
    try:
        a = ctx.scratch(("t", 1), 1000, np.float64)
        b = ctx.scratch(("t", 2), 1000, np.float64)
        assert a.ptr.value != b.ptr.value and abs(a.ptr.value - b.ptr.value) >= 8000
        assert ctx.scratch(("t", 1), 1000, np.float64).ptr.value == a.ptr.value
        chunks = ctx.__dict__["_arena_chunks"]
        assert len(chunks) == 1 and chunks[0][0].ptr.value <= a.ptr.value < chunks[0][0].ptr.value + chunks[0][0].n

        class Owner:
            pass
        o = Owner()
        c = ctx.scratch((ctx.owner_key(o), "x"), 5000, np.int16)
        where = c.ptr.value
        del o, c
        gc.collect()
        o2 = Owner()
        d = ctx.scratch((ctx.owner_key(o2), "x"), 5000, np.int16)
        assert d.ptr.value == where                            # the released range, same size: handed out again
        big = ctx.scratch(("t", "big"), (600 << 20), np.uint8)  # beyond a quarter of the block: its own allocation
        assert not (chunks[0][0].ptr.value <= big.ptr.value < chunks[0][0].ptr.value + chunks[0][0].n)
        x = np.arange(1000, dtype=np.float64)
        up = ctx.upload(x)
        from pymodem_amd._native import check, lib
        check(lib().pm_d2d(ctx.handle, a.ptr, up.ptr, x.nbytes))
        assert np.array_equal(ctx.scratch(("t", 1), 1000, np.float64).download(), x)
    finally:
        ctx.close()


def test_drop_scratch_gives_the_blocks_back_and_forgets_old_ranges():
    """Context.drop_scratch: pool and arena blocks go back to the device; a range handed out before the drop that is released
    afterwards (its owner collected late) is not put on the new free list, and later requests get fresh, working memory."""
    import gc
    import pymodem_amd
    ctx = pymodem_amd.Context.side(index=77, high_priority=False)

    class Owner:
        pass
    o = Owner()
    a = ctx.scratch((ctx.owner_key(o), "x"), 1000, np.float64)
    big = ctx.scratch(("big",), 80_000_000, np.float64)              # past the arena: a block of its own
    old_ptr = a.ptr.value
    assert ctx.__dict__.get("_arena_chunks")
    ctx.drop_scratch()
    assert not ctx.__dict__.get("_arena_chunks") and not ctx.__dict__.get("_pool")
    del o, a, big
    gc.collect()                                                     # the owner's finaliser runs now, with a range of the old block
    assert not ctx.__dict__.get("_arena_free")
    b = ctx.scratch(("y",), 1000, np.float64)
    x = np.arange(1000, dtype=np.float64)
    from pymodem_amd._native import check, lib
    import ctypes
    check(lib().pm_h2d(ctx.handle, b.ptr, x.ctypes.data_as(ctypes.c_void_p), x.nbytes))
    assert np.array_equal(b.download(), x) and old_ptr is not None


@pytest.mark.parametrize("cfg,rate", [("afsk_1200.json", 48000), ("fsk_9600.json", 48000), ("bpsk_300.json", 48000), ("qpsk_2400.json", 48000),
                                      ("afsk_300_pll.json", 8000), ("afsk_300.json", 8000)])
def test_whole_chain_entry_point_matches_the_stage_path(config_lines, cfg, rate):
    """pm_chain_create / pm_chain_run (modem + slicer in one C call, host or device audio) give the slicer bytes and addresses of
    the stage-by-stage path and of the oracle, for every modem family; a second run after pm_chain_reset repeats them."""
    from pymodem_amd import chain_builder as cb, chain_execute as ce
    import pymodem_amd
    ctx = pymodem_amd.Context.default()
    n = 120000 if rate == 48000 else 60000
    audio = noise_i16(n)
    for line in config_lines(cfg):
        chain = cb.build_chain(rate, line)
        want = chain[2].slice(chain[1].demod(audio, device_out=True))
        o = O.build_chain(rate, line)
        od, oa = o[1].slice(o[0].demod(audio, canon=True))
        assert np.array_equal(want.data, od) and np.array_equal(want.address, oa)
        fresh = cb.build_chain(rate, line)
        nc = ce.NativeChain(fresh[1], fresh[2])
        got = nc.run(audio)
        assert np.array_equal(got.data, od) and np.array_equal(got.address, oa), line["object_name"]
        nc.reset()
        again = nc.run(ctx.upload(audio))                 # audio already in HBM
        assert np.array_equal(again.data, od) and np.array_equal(again.address, oa)
        # the rest of the chain runs on the slicer output as usual
        p1 = fresh[4].decode(fresh[3].stream_unscramble_8bit(got))
        p2 = chain[4].decode(chain[3].stream_unscramble_8bit(want))
        assert [p.streamaddress for p in p1] == [p.streamaddress for p in p2]
        nc.close()


def test_whole_chain_entry_point_errors():
    from pymodem_amd import NativeError, chain_builder as cb, chain_execute as ce
    line = {"object_name": "x", "object_type": "demod_chain", "modem": {"type": "afsk", "config": "1200", "options": {}},
            "slicer": {"type": "binary", "config": "1200", "options": {}}, "stream": {"type": "lfsr", "options": {}}, "codec": {"type": "ax25"}}
    ch = cb.build_chain(48000, line)
    nc = ce.NativeChain(ch[1], ch[2])
    with pytest.raises(NativeError):
        nc.run(noise_i16(100))                            # shorter than the input filter
    assert len(nc.run(np.zeros(0, np.int16))) == 0


def test_whole_chain_capacity_error_keeps_the_run():
    """Buffers that are too small: PM_ERR_CAPACITY, the chain's state has advanced with the stream, and pm_chain_fetch hands the
    same run's output over again -- two pieces fed that way equal the single call on the whole input cut in the same place."""
    import ctypes
    from pymodem_amd import chain_builder as cb, chain_execute as ce
    from pymodem_amd._native import lib
    line = {"object_name": "x", "object_type": "demod_chain", "modem": {"type": "fsk", "config": "9600", "options": {}},
            "slicer": {"type": "binary", "config": "9600", "options": {}}, "stream": {"type": "lfsr", "options": {}}, "codec": {"type": "ax25"}}
    audio = noise_i16(96000)
    ch = cb.build_chain(48000, line)
    nc = ce.NativeChain(ch[1], ch[2])
    want = [nc.run(audio[:48000]), nc.run(audio[48000:])]
    nc.reset()
    got = []
    for piece in (audio[:48000], audio[48000:]):
        a = np.ascontiguousarray(piece)
        small_d, small_a, count = np.empty(16, np.uint8), np.empty(16, np.int64), ctypes.c_int64()
        rc = lib().pm_chain_run(nc._h, a.ctypes.data_as(ctypes.c_void_p), len(a), 0, small_d.ctypes.data_as(ctypes.c_void_p),
                                small_a.ctypes.data_as(ctypes.c_void_p), 16, ctypes.byref(count))
        assert rc == -4 and count.value > 16                      # PM_ERR_CAPACITY, *h_count = what is needed
        d, ad, c2 = np.empty(count.value, np.uint8), np.empty(count.value, np.int64), ctypes.c_int64()
        assert lib().pm_chain_fetch(nc._h, d.ctypes.data_as(ctypes.c_void_p), ad.ctypes.data_as(ctypes.c_void_p), count.value, ctypes.byref(c2)) == 0
        assert c2.value == count.value
        got.append((d, ad))
    for (d, ad), w in zip(got, want):
        assert np.array_equal(d, w.data) and np.array_equal(ad, w.address)
    nc.close()


def test_recording_pipeline_soak(config_lines):
    """Ninety recordings of three different kinds through one pipeline, several in flight on every stage: each result equals the
    one-at-a-time result for its kind (bitmap slots, slicer streams, host and post stages never mix recordings up)."""
    from pymodem_amd import chain_builder as cb, chain_execute as ce, siggen
    lines = config_lines("afsk_1200_ax25_super_opt.json")
    kinds = [siggen.recording("afsk1200_ax25", 48000, packets=4, seed=s, noise_sigma=500.0, payload_len=(20, 60))[0] for s in (1, 2, 3)]
    kinds[1] = kinds[1][: len(kinds[1]) // 4].copy()          # very different lengths: slicers finish out of submission order,
    kinds[2] = np.tile(kinds[2], 3)                           # so a slicer stream must belong to a task, not to a recording index
    want = [ce.process_chains_table([cb.build_chain(48000, l) for l in lines], a) for a in kinds]
    pipe = ce.RecordingPipeline()
    seen = []
    futures = []
    for k in range(90):
        which = (k * 7 + k // 5) % 3
        futures.append((which, pipe.submit([cb.build_chain(48000, l) for l in lines], kinds[which],
                                           finish=lambda rows, k=k: (seen.append(k), rows)[1],
                                           post=lambda rows: [r.copy() for r in rows])))
    for which, f in futures:
        rows = f.result()
        for ci in range(len(lines)):
            assert np.array_equal(rows[ci], want[which][ci]), (which, ci)
    pipe.close()
    assert seen == list(range(90))                                              # the ordered stage ran in submission order
    assert sum(len(r) for r in want[0].values()) > 0


def test_recording_pipeline_prefetch(config_lines):
    """Recordings that start in host memory: prefetch() copies the next one into HBM on a copy stream while the previous one is
    demodulated; three rotating device buffers must never be overwritten under a demod that still reads them."""
    from pymodem_amd import chain_builder as cb, chain_execute as ce, siggen
    lines = config_lines("afsk_1200_ax25_super_opt.json")
    kinds = [siggen.recording("afsk1200_ax25", 48000, packets=3, seed=s, noise_sigma=500.0, payload_len=(20, 60))[0] for s in (11, 12, 13, 14)]
    want = [ce.process_chains_table([cb.build_chain(48000, l) for l in lines], a) for a in kinds]
    pipe = ce.RecordingPipeline()
    order = [(k * 5 + k // 3) % 4 for k in range(40)]
    futures = []
    nxt = pipe.prefetch(kinds[order[0]])
    for i, which in enumerate(order):
        cur, nxt = nxt, (pipe.prefetch(kinds[order[i + 1]]) if i + 1 < len(order) else None)
        futures.append((which, pipe.submit([cb.build_chain(48000, l) for l in lines], cur)))
    for which, f in futures:
        rows = f.result()
        for ci in range(len(lines)):
            assert np.array_equal(rows[ci], want[which][ci]), (which, ci)
    pipe.close()


def test_gain_sweep_path_equals_exact_group_path(config_lines):
    """Group executor on afsk_1200_ax25_super_opt.json three ways -- certified signs from sliding correlator sums (default: the seven
    chains of the gain sweep in one entry, the eighth chain in an entry of its own), certified signs from the direct sums (the
    sweep only), and the exact correlator-group + batched low-pass path: identical slicer bytes, addresses and packets for every chain."""
    from pymodem_amd import chain_builder as cb, chain_execute as ce, siggen
    from pymodem_amd.modems import AFSKModem
    import pymodem_amd
    lines = config_lines("afsk_1200_ax25_super_opt.json")
    audio, _ = siggen.recording("afsk1200_ax25", 48000, packets=12, seed=21, noise_sigma=1500.0, payload_len=(20, 80))
    res = {}
    for name, sweep, sliding in (("sliding", True, True), ("direct", True, False), ("exact", False, True)):
        ce._USE_SWEEP, AFSKModem.sliding_sums = sweep, sliding
        before = AFSKModem.sweeps_run
        try:
            st = {}
            pk = ce.process_chains_device([cb.build_chain(48000, l) for l in lines], audio, stages=st)
            if sweep:
                assert 0 <= AFSKModem.sweep_uncertain(pymodem_amd.Context.default()) < 1000
        finally:
            ce._USE_SWEEP, AFSKModem.sliding_sums = True, True
        res[name] = (st["sliced"], pk, AFSKModem.sweeps_run - before)
    assert [res[k][2] for k in ("sliding", "direct", "exact")] == [2, 1, 0]          # which entries ran
    for other in ("direct", "exact"):
        for c in range(len(lines)):
            a, b = res["sliding"][0][c], res[other][0][c]
            assert np.array_equal(a.data, b.data) and np.array_equal(a.address, b.address), (other, c)
            assert [(p.streamaddress, bytes(bytearray(p.data))) for p in res["sliding"][1][c]] == \
                   [(p.streamaddress, bytes(bytearray(p.data))) for p in res[other][1][c]]


@pytest.mark.parametrize("kind", ["silence", "whisper"])
def test_deferred_sweep_fallback_on_degenerate_input(config_lines, kind, monkeypatch):
    """The group executor defers the certified sweeps' overflow fallback (pm_afsk_sweep_mode): on digital silence and on audio far below
    the stated bound every sample is uncertain, the list overflows, and the executor must redo those chains with the exact kernels --
    sequentially and in the pipelined executor.  Bytes and addresses equal the oracle's for every chain."""
    from pymodem_amd import chain_builder as cb, chain_execute as ce
    lines = config_lines("afsk_1200_ax25_super_opt.json")
    n = 400000
    audio = np.zeros(n, np.int16) if kind == "silence" else (np.random.default_rng(3).integers(-1, 2, n)).astype(np.int16)
    want = [O.run_chain(O.build_chain(48000, l), audio, canon=True) for l in lines]
    redone = []
    real = ce.resolve_sweeps
    monkeypatch.setattr(ce, "resolve_sweeps", lambda *a, **k: redone.append(real(*a, **k)) or redone[-1])
    st = {}
    ce.process_chains_device([cb.build_chain(48000, l) for l in lines], audio, stages=st)
    if kind == "silence":
        assert sum(redone) >= 1                                       # every sample uncertain: the lists overflowed, the chains were redone
    for c in range(len(lines)):
        assert np.array_equal(st["sliced"][c].data, want[c]["slice_data"]) and np.array_equal(st["sliced"][c].address, want[c]["slice_addr"]), c
    del redone[:]
    pipe = ce.RecordingPipeline()
    futs = [pipe.submit([cb.build_chain(48000, l) for l in lines], audio) for _ in range(4)]
    rows = [f.result() for f in futs]
    pipe.close()
    if kind == "silence":
        assert sum(redone) >= 4
    ref = ce.process_chains_table([cb.build_chain(48000, l) for l in lines], audio)
    for r in rows:
        for c in range(len(lines)):
            assert np.array_equal(r[c], ref[c])
    # The way bench.py drives it: ONE set of modem objects for every recording (reset in between), recordings uploaded a step ahead
    # (prefetch), degenerate and ordinary recordings interleaved, a dozen in flight.  The fallback runs on a slicer worker's thread
    # many recordings after the demod: it must neither disturb the modems the submitting thread is using nor read an upload buffer
    # that already holds a later recording.
    from pymodem_amd import siggen
    good = siggen.recording("afsk1200_ax25", 48000, packets=3, seed=5, noise_sigma=500.0, payload_len=(20, 60))[0][:n]
    quiet2 = np.zeros(len(good), np.int16) if kind == "silence" else (np.random.default_rng(4).integers(-1, 2, len(good))).astype(np.int16)
    kinds = [good, quiet2]
    refs = [ce.process_chains_table([cb.build_chain(48000, l) for l in lines], a) for a in kinds]
    modems = [cb.ModemConfigurator(48000, l["modem"]) for l in lines]

    def shared_chains():
        out = []
        for l, m in zip(lines, modems):
            m.reset()
            out.append([l["object_name"], m, cb.SlicerConfigurator(48000, l["slicer"]), cb.StreamConfigurator(l["stream"]),
                        cb.CodecConfigurator(l["codec"], l["object_name"])])
        return out
    del redone[:]
    pipe = ce.RecordingPipeline()
    order = [(k * 3 + k // 4) % 2 for k in range(24)]
    futs, nxt = [], pipe.prefetch(kinds[order[0]])
    for i, which in enumerate(order):
        cur, nxt = nxt, (pipe.prefetch(kinds[order[i + 1]]) if i + 1 < len(order) else None)
        futs.append((which, pipe.submit(shared_chains(), cur)))
    for which, f in futs:
        r = f.result()
        for c in range(len(lines)):
            assert np.array_equal(r[c], refs[which][c]), (which, c)
    pipe.close()
    if kind == "silence":
        assert sum(redone) >= order.count(1)


def test_bench_line_contract_with_and_without_the_exchange(tmp_path):
    """bench.py as the driver runs it: exactly one line on stdout, every field of the contract, the same packets whether the
    per-recording exchange really runs (one-rank RCCL gather, PYMODEM_AMD_FORCE_GATHER) or not."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = {}
    for name, extra in (("plain", {}), ("gather", {"PYMODEM_AMD_FORCE_GATHER": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533",
                                                    "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "12", "--warmup", "2", "--no-cpu-baseline", "--also", "0",
                            "--samples", "4800000"], capture_output=True, text=True, env=env, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        out = [ln for ln in r.stdout.split("\n") if ln.strip()]
        assert len(out) == 1, r.stdout[:500]                     # library banners must not reach stdout
        lines[name] = d = json.loads(out[0])
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                    "data", "config", "roofline"):
            assert key in d, key
        assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 2 and d["higher_is_better"] is True and d["value"] > 0
        assert d["config"]["workload"].startswith("afsk_1200_super_opt") and "model" not in d["config"]
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert key in d["roofline"], key
        assert d["roofline"]["bound"] in ("hbm", "mfma") and abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    assert lines["plain"]["packets"] == lines["gather"]["packets"] and lines["plain"]["packets"]["unique_good"] > 0
