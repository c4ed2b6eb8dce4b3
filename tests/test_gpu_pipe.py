"""The native pipelined executor (pm_pipe_*, pymodem_amd.chain_execute.NativePipeline) against the one-recording-at-a-time group
executor and the oracle: same packet rows per chain, same de-dup, for recordings of different lengths and kinds in flight together
(including the degenerate ones whose certified sweeps overflow and are redone with the exact kernels on the slicer worker)."""
import numpy as np
import pytest

from conftest import noise_i16
import oracle.oracle as O

pytestmark = pytest.mark.gpu

CFG = "afsk_1200_ax25_super_opt.json"


def _recordings():
    from pymodem_amd import siggen
    a = siggen.recording("afsk1200_ax25", 48000, packets=6, seed=11, noise_sigma=800.0, payload_len=(20, 80))[0]
    b = siggen.recording("afsk1200_ax25", 48000, packets=3, seed=12, noise_sigma=2500.0, payload_len=(10, 40))[0]
    gap = a.copy()                                            # squelch: stretches of digital silence inside a signal -- their workgroups certify
    gap[len(a) // 3: len(a) // 3 + 3000] = 0                  # nothing and hand everything to the sweep's list (decided on the slicer worker)
    gap[2 * len(a) // 3: 2 * len(a) // 3 + 700] = 0
    return {"a": a, "b": b, "half": a[: len(a) // 2].copy(), "noise": noise_i16(300000), "silence": np.zeros(250000, np.int16),
            "whisper": np.random.default_rng(3).integers(-1, 2, 200000).astype(np.int16), "short": a[:20000].copy(), "gap": gap}


def _want(lines, audio, rate=48000):
    from pymodem_amd import chain_builder as cb, chain_execute as ce
    from pymodem_amd.packet_meta import PacketTable
    rows = ce.process_chains_table([cb.build_chain(rate, l) for l in lines], audio)
    table = PacketTable(dict(rows), [l["object_name"] for l in lines]).correlate(rate / 40)
    return rows, table


@pytest.mark.parametrize("demod_streams", [1, 3])
def test_native_pipeline_matches_the_group_executor(config_lines, demod_streams):
    import pymodem_amd
    from pymodem_amd import chain_builder as cb, chain_execute as ce
    lines = config_lines(CFG)
    recs = _recordings()
    want = {k: _want(lines, v) for k, v in recs.items()}
    ctx = pymodem_amd.Context.default()
    dev = {k: ctx.upload(v) for k, v in recs.items()}
    ctx.sync()
    pipe = ce.NativePipeline([cb.build_chain(48000, l) for l in lines], max(len(v) for v in recs.values()), 48000 / 40, ctx=ctx,
                              demod_streams=demod_streams)
    order = ["a", "silence", "b", "half", "gap", "a", "noise", "whisper", "short", "b", "a", "silence", "gap", "half"] * 3
    tickets = [(k, pipe.submit(dev[k])) for k in order]
    tables = []
    for k, t in tickets:
        table = pipe.table(t)
        rows_w, table_w = want[k]
        at = 0
        for c in range(len(lines)):
            got = table.rows[at:at + table.counts[c]]
            at += table.counts[c]
            assert np.array_equal(got, rows_w[c]) or (
                len(got) == len(rows_w[c]) and all(np.array_equal(got[f], rows_w[c][f]) for f in got.dtype.names if f != "correlated_count")), (k, c)
        assert np.array_equal(table.unique_idx, table_w.unique_idx), k
        assert table.unique_decoders == table_w.unique_decoders, k
        assert table.CountGood() == table_w.CountGood() and table.CountBad() == table_w.CountBad()
        tables.append(table)
    assert want["a"][1].CountGood() >= 5
    st = pipe.stats()
    assert st["recordings"] == len(order) and st["slice_batches"] <= len(order)
    # a table outlives close(): its rows stay in the library until it is gone
    keep = tables[0].rows
    del tables, table
    pipe.close()
    assert np.array_equal(keep[: want["a"][1].counts[0]]["data"], want["a"][0][0]["data"])
    del keep


def test_released_rows_come_back_clean(config_lines):
    """A recording's packet rows go back to the pipeline's pool when its result is released, still written in, and are zeroed off the
    releasing thread (an idle host worker, or the next taker that finds no clean block: rows_retire / rows_get in pm_pipe.hip).  Whatever a
    later recording gets must hold zeros wherever its own packets did not write: whole rows -- the 1280-byte payload fields to their ends --
    against the group executor's, with results taken and released one at a time so that blocks are re-used while others wait to be cleaned,
    and recordings with many long packets followed by recordings with few short ones (or none)."""
    import pymodem_amd
    from pymodem_amd import chain_builder as cb, chain_execute as ce
    lines = config_lines(CFG)
    recs = _recordings()
    want = {k: _want(lines, v)[0] for k, v in recs.items() if k in ("a", "b", "half", "short", "silence")}
    ctx = pymodem_amd.Context.default()
    dev = {k: ctx.upload(recs[k]) for k in want}
    ctx.sync()
    pipe = ce.NativePipeline([cb.build_chain(48000, l) for l in lines], max(len(recs[k]) for k in want), 48000 / 40, ctx=ctx)
    order = ["a", "b", "a", "short", "half", "silence", "b", "a", "short", "b", "half", "a"] * 3
    tickets = [(k, pipe.submit(dev[k])) for k in order[:6]]
    nxt = 6
    seen = 0
    while tickets:
        k, t = tickets.pop(0)
        rows = pipe.rows(t)                                   # views of the library's memory: released when the last of them is gone
        for c in range(len(lines)):
            assert np.array_equal(rows[c], want[k][c]) or (
                len(rows[c]) == len(want[k][c]) and all(np.array_equal(rows[c][f], want[k][c][f]) for f in rows[c].dtype.names if f != "correlated_count")), (k, c, seen)
            seen += len(rows[c])
        del rows
        if nxt < len(order):
            tickets.append((order[nxt], pipe.submit(dev[order[nxt]])))
            nxt += 1
    assert seen > 100
    pipe.close()


def test_native_pipeline_rows_equal_the_oracle(config_lines):
    """Straight against the CPU restatement for one recording: slicer-to-packet results per chain."""
    import pymodem_amd
    from pymodem_amd import chain_builder as cb, chain_execute as ce, siggen
    lines = config_lines(CFG)
    audio = siggen.recording("afsk1200_ax25", 48000, packets=4, seed=31, noise_sigma=1200.0, payload_len=(20, 60))[0]
    ctx = pymodem_amd.Context.default()
    d = ctx.upload(audio)
    ctx.sync()
    pipe = ce.NativePipeline([cb.build_chain(48000, l) for l in lines], len(audio), 48000 / 40, ctx=ctx, chain_ids=[7 + c for c in range(len(lines))])
    tk = [pipe.submit(d) for _ in range(5)]
    for t in tk[:-1]:
        u, n = pipe.unique(t)
    table = pipe.table(tk[-1])
    assert (u, n) == (table.CountGood(), len(table.rows))
    at = 0
    for c, l in enumerate(lines):
        w = O.run_chain(O.build_chain(48000, l), audio, canon=True)
        got = table.rows[at:at + table.counts[c]]
        at += table.counts[c]
        assert [int(a) for a in got["streamaddress"]] == [int(p.streamaddress) for p in w["packets"]], c
        assert [bytes(r["data"][: r["len"]]) for r in got] == [bytes(bytearray(p.data)) for p in w["packets"]], c
        assert [int(b) for b in got["bytes_corrected"]] == [int(p.BytesCorrected) for p in w["packets"]], c
        assert (got["source_decoder"] == 7 + c).all()
    del table, got
    pipe.close()


def test_native_pipeline_refuses_what_it_does_not_cover(config_lines):
    import pymodem_amd
    from pymodem_amd import chain_builder as cb, chain_execute as ce
    ctx = pymodem_amd.Context.default()
    with pytest.raises(ValueError):
        ce.NativePipeline([cb.build_chain(48000, l) for l in config_lines("bpsk_300.json")], 100000, 1200.0, ctx=ctx)      # a carrier-loop modem
    lines = config_lines(CFG)
    pipe = ce.NativePipeline([cb.build_chain(48000, l) for l in lines], 100000, 1200.0, ctx=ctx)
    with pytest.raises(ValueError):
        pipe.submit(np.zeros(1000, np.int16))
    too_long = ctx.upload(np.zeros(100001, np.int16))
    with pytest.raises(pymodem_amd.NativeError):
        pipe.submit(too_long)
    tiny = ctx.upload(np.zeros(64, np.int16))
    with pytest.raises(pymodem_amd.NativeError):
        pipe.submit(tiny)
    with pytest.raises(pymodem_amd.NativeError):
        pipe.table(12345)
    ok = ctx.upload(noise_i16(100000))
    t = pipe.submit(ok)
    assert pipe.unique(t)[0] == 0
    pipe.close()


def test_submit_many_promises_its_tickets(config_lines):
    """pm_pipe_submit_many: a run's recordings from ONE library call on its own thread; tickets are announced before the first is
    submitted (pm_pipe_promise) and a wait on one of them waits for its submission.  Forty recordings through sixteen slots, the
    consumers started before the submitter, results equal to per-recording submits; a refused recording fails the waits behind it."""
    import threading
    import pymodem_amd
    from pymodem_amd import NativeError, chain_builder as cb, chain_execute as ce
    lines = config_lines(CFG)
    recs = _recordings()
    ctx = pymodem_amd.Context.default()
    order = ["a", "b", "half", "noise", "a", "short", "b", "a"] * 5
    dev = {k: ctx.upload(recs[k]) for k in set(order)}
    ctx.sync()
    pipe = ce.NativePipeline([cb.build_chain(48000, l) for l in lines], max(len(v) for v in recs.values()), 48000 / 40, ctx=ctx)
    one_by_one = {}
    for k in set(order):
        t = pipe.table(pipe.submit(dev[k]))
        one_by_one[k] = (t.counts, t.CountGood(), t.CountBad(), t.unique_idx.copy())
    first, join = pipe.submit_many([dev[k] for k in order])
    got, errors = {}, []

    def consume(lo, hi):
        try:
            for i in range(lo, hi):
                t = pipe.table(first + i)
                got[i] = (t.counts, t.CountGood(), t.CountBad(), t.unique_idx.copy())
        except BaseException as e:                            # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=consume, args=(lo, lo + 10)) for lo in range(0, 40, 10)]
    for th in reversed(threads):                              # the one waiting for the LAST tickets first
        th.start()
    join()
    for th in threads:
        th.join()
    assert not errors, errors
    for i, k in enumerate(order):
        w = one_by_one[k]
        assert got[i][:3] == w[:3] and np.array_equal(got[i][3], w[3]), (i, k)
    # a recording shorter than the filters stops the run: the submitter reports it, and the waits behind it do not hang
    tiny = ctx.upload(np.zeros(50, np.int16))
    first, join = pipe.submit_many([dev["a"], tiny, dev["a"]])
    assert pipe.table(first).CountGood() == one_by_one["a"][1]
    with pytest.raises(NativeError):
        join()
    with pytest.raises(NativeError):
        pipe.table(first + 2)
    t = pipe.table(pipe.submit(dev["b"]))                     # and the pipeline goes on
    assert t.CountGood() == one_by_one["b"][1]
    pipe.close()


@pytest.mark.parametrize("rate", [8000, 44100])
def test_native_pipeline_at_other_sample_rates(config_lines, rate):
    """The same comparison at 8 kHz and 44.1 kHz: other band-pass / correlator / low-pass lengths (the matrix-pipe plans are per tap
    set; 8 kHz leaves the band-pass with two dozen taps), other samples per symbol for the slicers."""
    import pymodem_amd
    from pymodem_amd import chain_builder as cb, chain_execute as ce, siggen
    lines = config_lines(CFG)
    recs = {"a": siggen.recording("afsk1200_ax25", rate, packets=5, seed=21, noise_sigma=900.0, payload_len=(20, 60))[0],
            "b": siggen.recording("afsk1200_ax25", rate, packets=2, seed=22, noise_sigma=2500.0, payload_len=(10, 30))[0],
            "noise": noise_i16(120001), "silence": np.zeros(90000, np.int16)}
    want = {k: _want(lines, v, rate) for k, v in recs.items()}
    ctx = pymodem_amd.Context.default()
    dev = {k: ctx.upload(v) for k, v in recs.items()}
    ctx.sync()
    pipe = ce.NativePipeline([cb.build_chain(rate, l) for l in lines], max(len(v) for v in recs.values()), rate / 40, ctx=ctx)
    order = ["a", "b", "noise", "a", "silence", "b"] * 2
    for k, t in [(k, pipe.submit(dev[k])) for k in order]:
        table = pipe.table(t)
        rows_w, table_w = want[k]
        at = 0
        for c in range(len(lines)):
            got = table.rows[at:at + table.counts[c]]
            at += table.counts[c]
            assert len(got) == len(rows_w[c]) and all(np.array_equal(got[f], rows_w[c][f]) for f in got.dtype.names if f != "correlated_count"), (k, c)
        assert np.array_equal(table.unique_idx, table_w.unique_idx) and table.unique_decoders == table_w.unique_decoders, k
    assert want["a"][1].CountGood() >= 3
    pipe.close()


@pytest.mark.parametrize("mixed", [False, True])
def test_native_pipeline_with_fsk_chains(config_lines, mixed):
    """Sign-FIR groups (pm_pipe_fir): the three chains of configs/fsk_9600.json share one FSK modem -- one pm_fir_signs_i16 per
    recording, every chain slices that bitmap -- alone and in one pipeline with the AFSK sweeps of the headline config; rows and
    de-dup equal to the group executor's."""
    import pymodem_amd
    from pymodem_amd import chain_builder as cb, chain_execute as ce, siggen
    lines = config_lines("fsk_9600.json") + (config_lines(CFG) if mixed else [])
    recs = {"f": siggen.recording("fsk9600_il2p", 48000, packets=6, seed=41, noise_sigma=900.0, payload_len=(20, 80))[0],
            "a": siggen.recording("afsk1200_ax25", 48000, packets=3, seed=42, noise_sigma=900.0, payload_len=(20, 40))[0],
            "noise": noise_i16(150001), "silence": np.zeros(100000, np.int16), "short": noise_i16(3000)}
    want = {k: _want(lines, v) for k, v in recs.items()}
    ctx = pymodem_amd.Context.default()
    dev = {k: ctx.upload(v) for k, v in recs.items()}
    ctx.sync()
    pipe = ce.NativePipeline([cb.build_chain(48000, l) for l in lines], max(len(v) for v in recs.values()), 48000 / 40, ctx=ctx)
    order = ["f", "a", "noise", "f", "silence", "short", "f", "a"] * 2
    for k, t in [(k, pipe.submit(dev[k])) for k in order]:
        table = pipe.table(t)
        rows_w, table_w = want[k]
        assert table.counts == table_w.counts, k
        at = 0
        for c in range(len(lines)):
            got = table.rows[at:at + table.counts[c]]
            at += table.counts[c]
            assert all(np.array_equal(got[f], rows_w[c][f]) for f in got.dtype.names if f != "correlated_count"), (k, c)
        assert np.array_equal(table.unique_idx, table_w.unique_idx) and table.unique_decoders == table_w.unique_decoders, k
    assert want["f"][1].CountGood() >= 4
    pipe.close()
