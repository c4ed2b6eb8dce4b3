"""The path bench.py times -- pm_pipe_* (NativePipeline): band-pass and low-pass sums as int8 digit products on the matrix pipe,
certified sign decisions, exact recomputation of the undecided ones, lockstep slicers, LFSR + codec on the library's threads --
pinned DIRECTLY at bitstream level (slicer.py:59-107, lfsr.py:22-52, chain_execute.py:30-52): for every chain the slicer's bytes,
their stream addresses, the LFSR's bytes and the packets against
  (a) the oracle (canonical FIR order) on the bench's own 28.8 M-sample buffer, all eight chains of the headline config, and
  (b) the reference's goldens (tests/golden/synth_chains.npz, 240 000 samples of seeded noise: the reference itself produced them)
      for afsk_1200_ax25_super_opt.json, afsk_1200.json and fsk_9600.json.
The pipeline keeps what its slicers and LFSRs produced when it is made with keep_slices (pm_pipe_slices): the arrays compared here
are the ones its codecs consumed, not a second computation."""
import numpy as np
import pytest

from conftest import noise_i16, oracle_chains
from oracle import oracle as O

pytestmark = pytest.mark.gpu
N = 28_800_000


def _pk(rows):
    return (rows["streamaddress"].astype(np.int64), rows["bytes_corrected"].astype(np.int64),
            np.concatenate([r["data"][: r["len"]] for r in rows]).astype(np.uint8) if len(rows) else np.zeros(0, np.uint8))


def _rows_by_chain(table, nchains):
    out, at = [], 0
    for c in range(nchains):
        out.append(table.rows[at:at + table.counts[c]])
        at += table.counts[c]
    return out


@pytest.mark.parametrize("cfg", ["afsk_1200_ax25_super_opt.json", "afsk_1200.json", "fsk_9600.json"])
def test_native_pipeline_bitstream_equals_the_reference_goldens(golden, config_lines, cfg):
    import pymodem_amd
    from pymodem_amd import chain_builder as cb, chain_execute as ce
    g = golden("synth_chains")
    lines = config_lines(cfg)
    audio = noise_i16(240000)
    ctx = pymodem_amd.Context.default()
    d = ctx.upload(audio)
    ctx.sync()
    pipe = ce.NativePipeline([cb.build_chain(48000, l) for l in lines], len(audio), 48000 / 40, ctx=ctx, keep_slices=True)
    tickets = [pipe.submit(d) for _ in range(3)]             # three in flight: both demod streams, one slicer batch
    for t in tickets:
        table = pipe.table(t)
        rows = _rows_by_chain(table, len(lines))
        for ci in range(len(lines)):
            prefix = f"{cfg[:-5]}__c{ci}__48k_l"
            sliced, plain = pipe.slices(t, ci)
            assert np.array_equal(sliced.data, g[prefix + "_slice_data"]), prefix
            assert np.array_equal(sliced.address, g[prefix + "_slice_addr"]), prefix
            assert np.array_equal(plain, g[prefix + "_lfsr_data"]), prefix
            a, c, dd = _pk(rows[ci])
            assert np.array_equal(a, g[prefix + "_pkt_addr"]) and np.array_equal(dd, g[prefix + "_pkt_data"]), prefix
            assert np.array_equal(c, g[prefix + "_pkt_corrected"]), prefix
        del table, rows
    pipe.close()


@pytest.mark.parametrize("workload,least_bytes,least_packets", [("afsk_1200_super_opt", 80000, 4000), ("fsk_9600", 700000, 1500)])
def test_native_pipeline_bitstream_at_full_size_equals_the_oracle(workload, least_bytes, least_packets):
    """The bench's buffer and config, every chain: what the timed path's slicers, LFSRs and codecs produce is what the CPU restatement
    of the reference produces.  Two recordings in flight (one per demod stream); the second is the buffer negated -- other
    packets' worth of bits, same statistics -- so that a result cannot come from the wrong slot.  The headline (configs[3]) and
    fsk_9600 (configs[2]: the short-tap sign FIR, three chains on one front end, IL2P and G3RUH AX.25 behind it)."""
    import bench
    import pymodem_amd
    from pymodem_amd import chain_builder as cb, chain_execute as ce

    class A:
        pass
    args = A()
    args.samples, args.rate, args.workload, args.buffer = N, 48000, workload, "signal"
    audio = bench.make_buffer(args)
    other = np.negative(np.maximum(audio, -32767))
    factory, cpg, _ = bench.WORKLOADS[args.workload]
    lines = [factory(c) for c in range(cpg)]
    ctx = pymodem_amd.Context.default()
    dev = [ctx.upload(audio), ctx.upload(other)]
    ctx.sync()
    pipe = ce.NativePipeline([cb.build_chain(48000, l) for l in lines], N, 48000 / 40, ctx=ctx, keep_slices=True)
    wanted = oracle_chains([(line, a) for a in (audio, other) for line in lines])
    tickets = [pipe.submit(b) for b in dev]
    total = 0
    for k, (t, a) in enumerate(zip(tickets, (audio, other))):
        table = pipe.table(t)
        rows = _rows_by_chain(table, len(lines))
        for c, line in enumerate(lines):
            w = wanted[k * len(lines) + c].result()
            sliced, plain = pipe.slices(t, c)
            assert np.array_equal(sliced.data, w["slice_data"]) and np.array_equal(sliced.address, w["slice_addr"]), c
            assert np.array_equal(plain, np.asarray(w["lfsr"], dtype=np.uint8)), c
            # ... and the demod stage on its own: every one of the 28.8 M sign bits the slicer read is the oracle's (round 5: pm_pipe_bitmap)
            want_bits = np.asarray(w["demod"]) >= 0
            got_bits = pipe.bitmap(t, c, len(want_bits))
            assert np.array_equal(got_bits, want_bits), (c, int(np.count_nonzero(got_bits != want_bits)))
            got_a, got_c, got_d = _pk(rows[c])
            assert [int(x) for x in got_a] == [int(p.streamaddress) for p in w["packets"]], c
            assert got_d.tobytes() == b"".join(bytes(bytearray(p.data)) for p in w["packets"]), c
            assert [int(x) for x in got_c] == [int(p.BytesCorrected) for p in w["packets"]], c
            total += len(w["packets"])
            assert len(sliced.data) > least_bytes
        del table, rows
    assert total > least_packets
    pipe.close()


def test_what_the_sweeps_leave_undecided_is_decided_by_the_reference_s_chain(config_lines):
    """A certified sweep on the matrix pipe decides its uncertain samples inside the workgroup that found them (the reference's chain
    from the int16 audio: afsk.py:151-166), hands a workgroup's overflow -- digital silence inside a signal: nothing there can be
    certified -- to its list, which the slicer worker works off, and a recording whose list overflows to the exact kernels.  All
    three against the oracle at bitstream level, every chain, the three kinds in flight together."""
    import pymodem_amd
    from pymodem_amd import chain_builder as cb, chain_execute as ce, siggen
    lines = config_lines("afsk_1200_ax25_super_opt.json")
    sig = siggen.recording("afsk1200_ax25", 48000, packets=5, seed=77, noise_sigma=1500.0, payload_len=(20, 60))[0][:330000]
    rng = np.random.default_rng(5)
    gap = sig.copy()
    gap[100000:104000] = 0                                    # two workgroups' worth of silence: ~28 000 list entries per sweep of seven
    gap[200000:200900] = 0
    quiet = sig.copy()
    quiet[50000:90000] = rng.integers(-2, 3, 40000)           # a stretch 70 dB down: scaled to itself, a few dozen uncertain samples per workgroup
    hush = sig.copy()
    hush[20000:300000] = 0                                    # more silence than the list holds: the exact kernels
    recs = {"signal": sig, "gap": gap, "quiet": quiet, "hush": hush}
    ctx = pymodem_amd.Context.default()
    dev = {k: ctx.upload(v) for k, v in recs.items()}
    ctx.sync()
    pipe = ce.NativePipeline([cb.build_chain(48000, l) for l in lines], len(sig), 48000 / 40, ctx=ctx, keep_slices=True)
    order = ["gap", "signal", "quiet", "hush", "gap", "quiet"]
    tickets = [(k, pipe.submit(dev[k])) for k in order]
    want = {k: [O.run_chain(O.build_chain(48000, l), v, canon=True) for l in lines] for k, v in recs.items()}
    for k, t in tickets:
        kept = [pipe.slices(t, c) for c in range(len(lines))]         # (before the table: a recording without packets is released with it)
        table = pipe.table(t)
        rows = _rows_by_chain(table, len(lines))
        for c in range(len(lines)):
            w = want[k][c]
            sliced, plain = kept[c]
            assert np.array_equal(sliced.data, w["slice_data"]) and np.array_equal(sliced.address, w["slice_addr"]), (k, c)
            assert np.array_equal(plain, np.asarray(w["lfsr"], dtype=np.uint8)), (k, c)
            got_a, _, got_d = _pk(rows[c])
            assert [int(x) for x in got_a] == [int(p.streamaddress) for p in w["packets"]], (k, c)
            assert got_d.tobytes() == b"".join(bytes(bytearray(p.data)) for p in w["packets"]), (k, c)
        del table, rows
    assert sum(len(w["packets"]) for w in want["signal"]) >= 8
    pipe.close()


@pytest.mark.parametrize("picked", [[0, 1, 3, 4], [2, 5], [0, 1, 2, 3, 4, 5]])
def test_native_pipeline_on_the_bundled_recording_equals_the_reference(golden, config_lines, picked):
    """The one recording the reference ships (afsk_300_il2pc_noise.wav, 8 kHz: a 187-tap band-pass, i.e. the FOUR-block band of the
    matrix-pipe band-pass) through the native executor with the AFSK correlator chains of configs/afsk_300_ax25.json, against what the
    reference itself produced (tests/golden/wav_chains.npz): slicer bytes, addresses, LFSR bytes, packets.  Four chains = two sweeps of
    two (one fused launch, both sweeps many-chain), two chains = one sweep (one fused launch), all six = three sweeps (the split path:
    band-pass + a launch per sweep)."""
    import os
    import pymodem_amd
    from conftest import GOLDEN, read_wav_pcm16
    from pymodem_amd import chain_builder as cb, chain_execute as ce
    g = golden("wav_chains")
    rate, audio = read_wav_pcm16(os.path.join(GOLDEN, "afsk_300_il2pc_noise.wav"))
    lines = config_lines("afsk_300_ax25.json")
    ctx = pymodem_amd.Context.default()
    d = ctx.upload(audio)
    ctx.sync()
    pipe = ce.NativePipeline([cb.build_chain(rate, lines[c]) for c in picked], len(audio), rate / 40, ctx=ctx, keep_slices=True)
    tickets = [pipe.submit(d) for _ in range(3)]
    for t in tickets:
        kept = [pipe.slices(t, k) for k in range(len(picked))]
        table = pipe.table(t)
        rows = _rows_by_chain(table, len(picked))
        for k, c in enumerate(picked):
            prefix = f"afsk_300_ax25__c{c}"
            sliced, plain = kept[k]
            assert np.array_equal(sliced.data, g[prefix + "_slice_data"]), prefix
            assert np.array_equal(sliced.address, g[prefix + "_slice_addr"]), prefix
            assert np.array_equal(plain, g[prefix + "_lfsr_data"]), prefix
            a, corr, data = _pk(rows[k])
            assert np.array_equal(a, g[prefix + "_pkt_addr"]) and np.array_equal(data, g[prefix + "_pkt_data"]), prefix
        del table, rows
    pipe.close()


def test_fused_launch_at_the_edges_of_its_tiles(config_lines):
    """Recording lengths that put the last output of a sweep on, just before and just after the boundaries the fused kernel works in --
    a bitmap word (64), a matrix tile (256), a workgroup (2048) -- for BOTH sweeps of the headline config at once (their correlators
    are 40 and 60 taps long: the seven-chain sweep has 20 outputs fewer, so one of them ends a word / tile / workgroup where the other
    does not), down to a single output.  The SIGN BITMAP of every chain, bit for bit, against the oracle's demodulated stream (and
    nothing set past its last bit), and the slicer's bytes and addresses; several lengths in flight."""
    import pymodem_amd
    from pymodem_amd import chain_builder as cb, chain_execute as ce
    lines = config_lines("afsk_1200_ax25_super_opt.json")
    chains = [cb.build_chain(48000, l) for l in lines]
    mb, ml = len(chains[0][1].input_bpf), len(chains[0][1].output_lpf)
    m0 = max(len(ch[1].mark_correlator_i) for ch in chains)   # the longest correlator: its sweep has the fewest outputs
    least = mb + m0 + ml - 2                                   # samples for ONE output of that sweep
    rng = np.random.default_rng(2048)
    base = np.clip(np.rint(rng.standard_normal(least + 3 * 2048 + 200) * 6000), -32768, 32767).astype(np.int16)
    outs = [1, 2, 20, 21, 63, 64, 65, 84, 85, 255, 256, 257, 276, 2047, 2048, 2049, 2068, 2069, 4096, 4097, 6143, 6144, 6165]
    ctx = pymodem_amd.Context.default()
    pipe = ce.NativePipeline(chains, len(base), 48000 / 40, ctx=ctx, keep_slices=True)
    dev = {k: ctx.upload(base[: least + k - 1].copy()) for k in outs}
    ctx.sync()
    for g0 in range(0, len(outs), 8):                         # (eight in flight: a bitmap slot is reused sixteen submissions later)
        for k, t in [(k, pipe.submit(dev[k])) for k in outs[g0:g0 + 8]]:
            audio = base[: least + k - 1]
            for c, line in enumerate(lines):
                w = O.run_chain(O.build_chain(48000, line), audio, canon=True)
                sliced, _ = pipe.slices(t, c)
                assert np.array_equal(sliced.data, w["slice_data"]) and np.array_equal(sliced.address, w["slice_addr"]), (k, c)
                want = np.asarray(w["demod"]) >= 0
                words = (len(want) + 63) // 64 * 64
                got = pipe.bitmap(t, c, words)
                assert np.array_equal(got[: len(want)], want), (k, c, int(np.count_nonzero(got[: len(want)] != want)))
                assert not got[len(want):].any(), (k, c)         # bits past the last output of the last word are zero
            pipe.release(t)
    pipe.close()


def test_a_refused_recording_leaves_no_hole_in_the_tickets(config_lines):
    """pm_pipe_submit refuses a recording that is long enough for the band-pass but not for a chain's correlator + low-pass
    (mb <= n < mb + m + ml - 2) before a ticket exists; the submit_many behind it gets consecutive tickets and the wait on its LAST
    ticket returns (it used to compare the ticket with a count of submissions and hang)."""
    import threading
    import pymodem_amd
    from pymodem_amd import NativeError, chain_builder as cb, chain_execute as ce
    lines = config_lines("afsk_1200_ax25_super_opt.json")
    ctx = pymodem_amd.Context.default()
    chains = [cb.build_chain(48000, l) for l in lines]
    mb = len(chains[0][1].input_bpf)
    ok = ctx.upload(noise_i16(120000))
    starved = ctx.upload(noise_i16(mb + 20))                 # passes the band-pass, starves the correlators and the low-pass
    ctx.sync()
    pipe = ce.NativePipeline(chains, 120000, 48000 / 40, ctx=ctx)
    first = pipe.submit(ok)
    with pytest.raises(NativeError):
        pipe.submit(starved)
    start, join = pipe.submit_many([ok] * 5)
    assert start == first + 1                                # the refused recording took no ticket
    got = []
    waiter = threading.Thread(target=lambda: got.append(pipe.unique(start + 4)))
    waiter.start()
    join()
    waiter.join(60)
    assert not waiter.is_alive() and got, "pm_pipe_wait on the last promised ticket did not return"
    for t in [first] + list(range(start, start + 4)):
        assert pipe.unique(t) == got[0]
    pipe.close()


@pytest.mark.parametrize("cfg,rate", [("afsk_1200_ax25_super_opt.json", 44100), ("afsk_1200_ax25_super_opt.json", 96000), ("afsk_1200_ax25_super_opt.json", 22050),
                                      ("afsk_1200.json", 44100), ("afsk_1200.json", 12000), ("afsk_300_ax25.json", 11025)])
def test_native_pipeline_at_other_sample_rates_equals_the_oracle(config_lines, cfg, rate):
    """Every filter of an AFSK chain is designed for the recording's sample rate (afsk.py:39-147: band-pass, correlator and low-pass
    lengths all scale with it), so other rates are other kernels' worth of shapes: tap counts that are not multiples of anything, runs of
    digits of other lengths, groups that do or do not qualify for the fused launch (pm_afsk_group_run_plan decides; the split launches
    take the rest).  Signal and noise, two recordings in flight, every chain at bitstream level and bit by bit."""
    import pymodem_amd
    from pymodem_amd import chain_builder as cb, chain_execute as ce, siggen
    lines = config_lines(cfg)
    mode = "afsk300_ax25" if cfg.startswith("afsk_300") else "afsk1200_ax25"
    sig = siggen.recording(mode, rate, packets=3, seed=rate % 97, noise_sigma=900.0, payload_len=(20, 50))[0]
    n = min(len(sig), 40 * rate // 10)
    a0 = np.ascontiguousarray(sig[:n])
    a1 = noise_i16(n, seed=rate % 89, sigma=5000.0)
    ctx = pymodem_amd.Context.default()
    dev = [ctx.upload(a0), ctx.upload(a1)]
    ctx.sync()
    pipe = ce.NativePipeline([cb.build_chain(rate, l) for l in lines], n, rate / 40, ctx=ctx, keep_slices=True)
    tickets = [pipe.submit(b) for b in dev]
    found = 0
    for t, a in zip(tickets, (a0, a1)):
        want = [O.run_chain(O.build_chain(rate, line), a, canon=True) for line in lines]
        for c, w in enumerate(want):                          # (slices and bitmaps first: a table without rows gives its ticket back)
            sliced, plain = pipe.slices(t, c)
            want_bits = np.asarray(w["demod"]) >= 0
            got_bits = pipe.bitmap(t, c, len(want_bits))
            assert np.array_equal(got_bits, want_bits), (cfg, rate, c, int(np.count_nonzero(got_bits != want_bits)))
            assert np.array_equal(sliced.data, w["slice_data"]) and np.array_equal(sliced.address, w["slice_addr"]), (cfg, rate, c)
            assert np.array_equal(plain, np.asarray(w["lfsr"], dtype=np.uint8)), (cfg, rate, c)
        table = pipe.table(t)
        rows = _rows_by_chain(table, len(lines))
        for c, w in enumerate(want):
            got_a, got_c, got_d = _pk(rows[c])
            assert [int(x) for x in got_a] == [int(p.streamaddress) for p in w["packets"]], (cfg, rate, c)
            assert got_d.tobytes() == b"".join(bytes(bytearray(p.data)) for p in w["packets"]), (cfg, rate, c)
            found += len(w["packets"])
        del table, rows
    pipe.close()
    assert found >= 1 or rate < 20000, (cfg, rate, found)
