"""The batch engine for the carrier-loop modems (pm_lbatch, pymodem_amd/loop_batch.py) and the rows kernels it is made of, against
the per-recording path (itself bit-exact against the oracle: tests/test_gpu_chains.py) and against the oracle directly.  The engine
only changes WHEN a statement runs (time chunks, many recordings per launch), so everything is compared bit for bit."""
import ctypes
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, noise_i16, tuned
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def ctx_lib():
    import pymodem_amd
    from pymodem_amd._native import check, lib
    return pymodem_amd.Context.default(), lib(), check


def lines_of(name):
    with open(os.path.join(GOLDEN, "configs", name)) as f:
        return [l for l in (json.loads(s) for s in f if s.strip()) if l.get("object_type") == "demod_chain"]


def bits_of(buf, n):
    words = buf.download((n + 63) // 64)
    return np.unpackbits(words.view(np.uint8), bitorder="little")[:n]


# ---- rows kernels -----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m", [1, 8, 130, 163, 241, 961])
def test_fir_rows_equal_the_single_stream_kernels(m):
    ctx, L, check = ctx_lib()
    rng = np.random.default_rng(m)
    rows, n, stride = 5, 9000 + m, 9000 + m + 24
    h = rng.standard_normal(m)
    dh = ctx.upload(h)
    xi = noise_i16(rows * stride, seed=m).reshape(rows, stride)
    xf = rng.standard_normal((rows, stride)) * 100
    nout = n - m + 1
    ystride = (nout + 9) // 2 * 2
    d_xi, d_xf = ctx.upload(xi.reshape(-1)), ctx.upload(xf.reshape(-1))
    d_y = ctx.empty(rows * ystride, np.float64)
    one = ctx.empty(nout, np.float64)
    for neg in (0, 1):
        # int16 rows (strided)
        check(L.pm_fir_rows_i16(ctx.handle, d_xi.ptr, stride, rows, n, dh.ptr, m, d_y.ptr, ystride, neg))
        got = d_y.download().reshape(rows, ystride)[:, :nout]
        for r in range(rows):
            check(L.pm_fir_valid_i16(ctx.handle, d_xi.view(r * stride, n).ptr, n, dh.ptr, m, one.ptr, neg))
            assert np.array_equal(got[r], one.download()), (m, r, "i16")
        want = O.fir_canon(xi[0, :n], h)
        assert np.array_equal(got[0], -want if neg else want)
        # float64 rows
        check(L.pm_fir_rows_f64(ctx.handle, d_xf.ptr, stride, rows, n, dh.ptr, m, d_y.ptr, ystride, neg))
        got = d_y.download().reshape(rows, ystride)[:, :nout]
        for r in range(rows):
            check(L.pm_fir_valid_f64(ctx.handle, d_xf.view(r * stride, n).ptr, n, dh.ptr, m, one.ptr, neg))
            assert np.array_equal(got[r], one.download()), (m, r, "f64")
    # sign-only rows, written at a word offset inside a larger bitmap array (what a chunk of the engine does)
    bstride = (nout + 63) // 64 + 3
    d_bits = ctx.upload(np.zeros(rows * bstride + 2, np.uint64))
    check(L.pm_fir_rows_signs_f64(ctx.handle, d_xf.ptr, stride, rows, n, dh.ptr, m, d_bits.view(2, rows * bstride).ptr, bstride, 0))
    words = d_bits.download()
    assert not words[:2].any()
    for r in range(rows):
        got = np.unpackbits(words[2 + r * bstride:2 + (r + 1) * bstride].view(np.uint8), bitorder="little")[:nout]
        want = O.fir_canon(xf[r, :n], h) >= 0
        assert np.array_equal(got.astype(bool), want), (m, r, "signs")


def test_fir_rows_from_a_pointer_table_at_any_offset():
    ctx, L, check = ctx_lib()
    rng = np.random.default_rng(5)
    m, n = 130, 20000
    h = rng.standard_normal(m)
    dh = ctx.upload(h)
    recs = [noise_i16(n + 100, seed=k) for k in range(3)]
    bufs = [ctx.upload(r) for r in recs]
    table = ctx.upload(np.array([b.ptr.value for b in bufs], dtype=np.uint64))
    nout = n - m + 1
    ystride = nout + (nout & 1) + 6
    d_y = ctx.empty(3 * ystride, np.float64)
    for off, aligned in [(0, 1), (8, 1), (3, 0), (37, 0)]:
        check(L.pm_fir_rows_i16_ptrs(ctx.handle, table.ptr, off, aligned, 3, n, dh.ptr, m, d_y.ptr, ystride, 0))
        got = d_y.download().reshape(3, ystride)[:, :nout]
        for r in range(3):
            assert np.array_equal(got[r], O.fir_canon(recs[r][off:off + n], h)), (off, r)


def test_agc_rows_in_pieces_equal_agc_apply_on_the_whole():
    from pymodem_amd._native import AGCParams
    ctx, L, check = ctx_lib()
    rows, n = 4, 50000
    rng = np.random.default_rng(11)
    t = np.arange(n) / 48000.0
    x = np.stack([np.sin(2 * np.pi * (1200 + 100 * r) * t) * (200 + 150 * np.sin(2 * np.pi * 0.7 * t + r)) * (1 + (t > 0.5)) + rng.standard_normal(n) * 5
                  for r in range(rows)])
    x[2, 30000:] = 0.0                                                   # the envelope decays to zero and the division is skipped
    p = AGCParams(500.0, 50.0, 0.2, 48000.0, 1.0)
    want, wstate = [], []
    for r in range(rows):
        buf = ctx.upload(x[r])
        st = (ctypes.c_double * 2)(0.0, 0.0)
        check(L.pm_agc_apply(ctx.handle, buf.ptr, n, ctypes.byref(p), st))
        want.append(buf.download())
        wstate.append((st[0], st[1]))
        ref, _ = O.agc_apply(x[r].copy(), 48000.0, 500.0, 0.2, 50.0, 1.0)
        assert np.array_equal(want[-1], ref)
    stride = n + 8
    d_x = ctx.upload(np.pad(x, ((0, 0), (0, 8))).reshape(-1))
    d_y = ctx.empty(rows * stride, np.float64)
    mx = (ctypes.c_double * rows)()
    check(L.pm_rows_max_f64(ctx.handle, d_x.ptr, stride, rows, n, mx))
    assert list(mx) == [x[r].max() for r in range(rows)]
    state = (ctypes.c_double * (2 * rows))()
    at = 0
    for piece in (1, 255, 256, 257, 4096, 20000, n):                       # uneven pieces, the last one takes the rest
        cnt = min(piece, n - at)
        check(L.pm_agc_rows_apply(ctx.handle, d_x.view(at, rows * stride - at).ptr, stride, d_y.view(at, rows * stride - at).ptr, stride, rows, cnt,
                                  ctypes.byref(p), mx, state))
        at += cnt
    assert at == n
    got = d_y.download().reshape(rows, stride)[:, :n]
    for r in range(rows):
        assert np.array_equal(got[r], want[r]), r
        assert (state[2 * r], state[2 * r + 1]) == wstate[r]


# ---- the engine against the stage objects -----------------------------------------------------------------------------------------
def group_modems(cfg, rate, carriers=None, take=None):
    from pymodem_amd import chain_builder as cb
    out = []
    for line in lines_of(cfg)[:take]:
        if carriers is None:
            out.append((line, cb.ModemConfigurator(rate, line["modem"])))
        else:
            for f in carriers:
                ln = json.loads(json.dumps(line))
                ln["modem"]["options"]["carrier_freq"] = str(f)
                ln["object_name"] += f" {f}"
                out.append((ln, cb.ModemConfigurator(rate, ln["modem"])))
            break
    return out


@pytest.mark.parametrize("cfg,rate,carriers,chunk", [
    ("bpsk_300.json", 48000, [1500.0], 2048),
    ("bpsk_300.json", 48000, [1490.0, 1500.0, 1512.5], 4096),
    ("bpsk_1200.json", 8000, [1500.0, 1510.0], 2048),
    ("qpsk_2400.json", 48000, [1475.0, 1500.0, 1525.0], 2048),
    ("qpsk_2400.json", 48000, [1500.0 + 3.125 * k for k in range(-4, 4)], 6144),
    ("qpsk_600.json", 44100, [1500.0], 0),
    ("afsk_300_pll.json", 8000, None, 2048),
])
@pytest.mark.parametrize("wide", [0, 1, 2])
def test_engine_bitmaps_equal_demod_signs(cfg, rate, carriers, chunk, wide):
    """Every (recording, chain) bitmap of a run equals modem.demod_signs() on that recording: different audio per recording, chunk
    lengths from one FIR tile up, chains per recording that do and do not divide the loops of a wave.  wide: the engine's loop
    launches in the shape for runs of thousands of loops (every lane of the stepping wave a loop, 32-sample tiles: PM_LOOP_WIDE),
    the per-recording reference in the eight-loop shape."""
    import pymodem_amd
    from pymodem_amd.loop_batch import LoopBatch
    ctx = pymodem_amd.Context.default()
    group = group_modems(cfg, rate, carriers, take=1 if carriers is None else None)
    modems = [m for _, m in group]
    n = 30000 if rate >= 44100 else 12000
    recs = [noise_i16(n, seed=100 + k, sigma=3000.0 + 2500.0 * k) for k in range(5)]
    recs[3][n // 2:] //= 16                                             # a level step: the AGC's sustain and decay at work
    eng = LoopBatch(modems, recordings=6, ctx=ctx, chunk=chunk)
    try:
        for take in (5, 2):                                             # a second run on the same engine starts from fresh states
            dev = [ctx.upload(r) for r in recs[:take]]
            with tuned(ctx, loop_wide=wide):
                got = eng.run(dev)
                ctx.sync()
            ctx.tune(loop_wide=0)
            for k in range(take):
                for c, (line, _) in enumerate(group):
                    from pymodem_amd import chain_builder as cb
                    ref = cb.ModemConfigurator(rate, line["modem"]).demod_signs(recs[k])
                    assert got[k][c].n == ref.n, (cfg, k, c)
                    assert np.array_equal(bits_of(got[k][c].bits_i, ref.n), bits_of(ref.bits_i, ref.n)), (cfg, k, c, "I")
                    if ref.bits_q is not None:
                        assert np.array_equal(bits_of(got[k][c].bits_q, ref.n), bits_of(ref.bits_q, ref.n)), (cfg, k, c, "Q")
    finally:
        ctx.tune(loop_wide=-1)
        eng.close()


@pytest.mark.parametrize("n", [5, 5003, 5008])
def test_loop_kernel_shapes_agree(n):
    """150 loops in one launch (three workgroups of the 64-loop shape, the last one part full), a length that is no multiple of either
    tile: outputs and end states of the four loop kernels bit for bit the same in both shapes, own input rows and one shared row."""
    import ctypes
    import math
    import pymodem_amd
    from pymodem_amd import taps as T
    from pymodem_amd._native import Loop, check, lib
    ctx = pymodem_amd.Context.default()
    nl = 150
    tab = ctx.upload(np.array([math.sin(i * 2.0 * math.pi / 256) for i in range(256)]))
    pd = ctx.upload(np.ascontiguousarray(T.qpsk_error_table().reshape(-1), dtype=np.int32))
    b0, b1, a1 = T.one_pole_lowpass(48000.0, 250.0, 1.0)
    rng = np.random.default_rng(9)
    x_own = ctx.upload(rng.standard_normal(n * nl) * 0.6)
    x_im = ctx.upload(rng.standard_normal(n * nl) * 0.6)

    def fresh():
        loops = (Loop * nl)()
        for k in range(nl):
            lp = loops[k]
            lp.phase_scaling, lp.index_scaling, lp.set_frequency = 2.0 * math.pi / 48000.0, 256 / (2.0 * math.pi), 1400.0 + 1.7 * k
            lp.b0, lp.b1, lp.a1 = b0, b1, a1
            lp.bb0, lp.bb1, lp.ba1 = b0, b1, a1
            lp.p_rate, lp.i_rate, lp.i_limit, lp.gain = 0.3, 0.3 / 2000, 31.25, 14400 / 65536
        return loops
    L = lib()
    o1, o2 = ctx.empty(n * nl, np.float64), ctx.empty(n * nl, np.float64)
    calls = {
        "costas own rows": lambda lp: check(L.pm_costas_bpsk(ctx.handle, lp, nl, tab.ptr, x_own.ptr, n, n, o1.ptr, n)),
        "costas shared": lambda lp: check(L.pm_costas_bpsk(ctx.handle, lp, nl, tab.ptr, x_own.ptr, 0, n, o1.ptr, n)),
        "pll own rows": lambda lp: check(L.pm_pll_afsk(ctx.handle, lp, nl, tab.ptr, x_own.ptr, n, n, o1.ptr, n)),
        "qpsk costas own rows": lambda lp: check(L.pm_costas_qpsk(ctx.handle, lp, nl, tab.ptr, x_own.ptr, n, n, o1.ptr, o2.ptr, n)),
        "mpsk own rows": lambda lp: check(L.pm_mpsk_loop(ctx.handle, lp, nl, tab.ptr, pd.ptr, x_own.ptr, x_im.ptr, n, n, o1.ptr, o2.ptr, n)),
        "mpsk shared": lambda lp: check(L.pm_mpsk_loop(ctx.handle, lp, nl, tab.ptr, pd.ptr, x_own.ptr, x_im.ptr, 0, n, o1.ptr, o2.ptr, n)),
    }
    for name, call in calls.items():
        got = {}
        for wide in (0, 1, 2):
            ctx.tune(loop_wide=wide)
            check(lib().pm_memset(ctx.handle, o1.ptr, 0, n * nl * 8))
            check(lib().pm_memset(ctx.handle, o2.ptr, 0, n * nl * 8))
            lp = fresh()
            call(lp)
            got[wide] = (o1.download(), o2.download(), bytes(lp))
        assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1]), name
        assert np.array_equal(got[0][0], got[2][0]) and np.array_equal(got[0][1], got[2][1]) and got[0][2] == got[2][2], (name, "direct")
        assert got[0][2] == got[1][2], name
        assert np.any(got[0][0] != 0), name
    ctx.tune(loop_wide=-1)


@pytest.mark.parametrize("cfg,rate,carriers,switches", [
    ("bpsk_300.json", 48000, [1500.0], dict(lbatch_loop_cus=16, loop_wide=2)),                 # loops on 16 units of their own, AGC in the loop's lane
    ("bpsk_300.json", 48000, [1500.0], dict(lbatch_loop_cus=0, loop_wide=2, loop_agc=0)),       # the AGC as a pass of its own again
    ("bpsk_300.json", 48000, [1500.0], dict(fir8=0, loop_wide=2)),                              # the matched filter in binary64 on the vector pipe
    ("bpsk_300.json", 48000, [1500.0], dict(bpf8_max=0, loop_wide=2)),                          # the AGC's `normal` from the reference's sums, chunk by chunk
    ("qpsk_2400.json", 48000, [1475.0, 1500.0, 1525.0], dict(bpf8_max=0, loop_wide=2)),
    ("qpsk_2400.json", 48000, [1475.0, 1500.0, 1525.0], dict(lbatch_loop_cus=32, loop_wide=2)),  # two-output loops, tiles, own units
    ("qpsk_2400.json", 48000, [1475.0, 1500.0, 1525.0], dict(loop_wide=2, loop_vec=0)),          # plain eight-byte stores
])
def test_engine_shapes_of_round_4_agree(cfg, rate, carriers, switches):
    """What round 4 added to the engine -- compute units reserved for the loops (CU-masked streams), the BPSK AGC stepped in the
    loop's lane, two-output loops storing through transposing tiles, matched filters on the matrix pipe -- each switched on and off
    (pm_ctx_tune): every bitmap still equals modem.demod_signs() on that recording."""
    import pymodem_amd
    from pymodem_amd import chain_builder as cb
    from pymodem_amd.loop_batch import LoopBatch
    ctx = pymodem_amd.Context.default()
    group = group_modems(cfg, rate, carriers)
    n = 30000
    recs = [noise_i16(n, seed=300 + k, sigma=2500.0 + 3000.0 * k) for k in range(4)]
    recs[2][n // 3:] //= 8
    def reference(line, audio):                               # (its bitmaps live in pooled work buffers: read them before the next call)
        r = cb.ModemConfigurator(rate, line["modem"]).demod_signs(audio)
        return r.n, bits_of(r.bits_i, r.n), None if r.bits_q is None else bits_of(r.bits_q, r.n)
    want = [[reference(line, recs[k]) for line, _ in group] for k in range(len(recs))]
    with tuned(ctx, **switches):
        eng = LoopBatch([m for _, m in group], recordings=4, ctx=ctx, chunk=4096)
        try:
            assert (eng.loop is not None) == (switches.get("lbatch_loop_cus", 0) > 0)
            got = eng.run([ctx.upload(r) for r in recs])
            ctx.sync()
            for k in range(len(recs)):
                for c in range(len(group)):
                    n_ref, bi, bq = want[k][c]
                    assert got[k][c].n == n_ref
                    assert np.array_equal(bits_of(got[k][c].bits_i, n_ref), bi), (cfg, switches, k, c, "I")
                    if bq is not None:
                        assert np.array_equal(bits_of(got[k][c].bits_q, n_ref), bq), (cfg, switches, k, c, "Q")
        finally:
            eng.close()


def test_context_switches_are_named():
    import pymodem_amd
    ctx = pymodem_amd.Context.default()
    ctx.tune(slicer_trace=0)
    with pytest.raises(pymodem_amd.NativeError):
        ctx.tune(no_such_switch=1)


def test_engine_qpsk_modem():
    """QPSKModem (psk.py:197-476, chain_builder type 'qpsk'): one input, two low-passed arms."""
    import pymodem_amd
    from pymodem_amd import chain_builder as cb
    from pymodem_amd.loop_batch import LoopBatch
    ctx = pymodem_amd.Context.default()
    spec = {"type": "qpsk", "config": "2400", "options": {}}
    modems = [cb.ModemConfigurator(48000, spec)]
    recs = [noise_i16(20000, seed=7 + k) for k in range(3)]
    eng = LoopBatch(modems, recordings=3, ctx=ctx, chunk=4096)
    try:
        got = eng.run([ctx.upload(r) for r in recs])
        ctx.sync()
        for k in range(3):
            ref = cb.ModemConfigurator(48000, spec).demod_signs(recs[k])
            assert np.array_equal(bits_of(got[k][0].bits_i, ref.n), bits_of(ref.bits_i, ref.n))
            assert np.array_equal(bits_of(got[k][0].bits_q, ref.n), bits_of(ref.bits_q, ref.n))
    finally:
        eng.close()


@pytest.mark.parametrize("cfg,rate,carriers,chunk,n", [
    ("bpsk_300.json", 48000, [1500.0], 2048, 30000),
    ("bpsk_300.json", 48000, [1490.0, 1500.0, 1512.5], 4096, 41234),
    ("bpsk_1200.json", 8000, [1500.0, 1510.0], 2048, 12000),
    ("qpsk_2400.json", 48000, [1500.0 + 3.125 * k for k in range(-4, 4)], 6144, 30001),
    ("qpsk_600.json", 44100, [1500.0], 0, 30000),
    ("afsk_300_pll.json", 8000, None, 2048, 12000),
])
def test_sliced_run_equals_run_then_slice_batch(cfg, rate, carriers, chunk, n):
    """pm_lbatch_run_sliced -- the slicers inside the engine, one lane per stream, state carried from chunk to chunk -- against the same
    engine's bitmaps sliced by pm_slice_batch (itself pinned to the oracle's slicers in tests/test_gpu_slicer.py): bytes, addresses and
    the slicer objects' end states, for binary and quadrature slicers, several chunks and a ragged last word, noise (a crossing
    at nearly every sample) and level steps."""
    import pymodem_amd
    from pymodem_amd import chain_builder as cb
    from pymodem_amd.loop_batch import LoopBatch
    from pymodem_amd.slicer import slice_batch
    ctx = pymodem_amd.Context.default()
    group = group_modems(cfg, rate, carriers, take=1 if carriers is None else None)
    modems = [m for _, m in group]
    recs = [noise_i16(n, seed=500 + k, sigma=2000.0 + 2500.0 * k) for k in range(5)]
    recs[1][n // 2:] //= 16
    t = np.arange(n)
    recs[2] = (6000 * np.sin(2 * np.pi * 1500.0 * t / rate) * np.sign(np.sin(2 * np.pi * 37.0 * t / rate))).astype(np.int16)      # a keyed carrier
    def slicers():
        return [cb.build_chain(rate, line)[2] for line, _ in group]
    eng = LoopBatch(modems, recordings=5, ctx=ctx, chunk=chunk)
    try:
        dev = [ctx.upload(r) for r in recs]
        run = eng.run_sliced(dev, slicers())
        ctx.sync()
        assert run.ok()
        got = run.fetch(0, run.rows)
        bitmaps = eng.run(dev)
        ctx.sync()
        C = len(group)
        assert run.rows == 5 * C and run.nout == bitmaps[0][0].n
        produced = 0
        for k in range(5):
            sls = slicers()
            want = slice_batch(sls, [sl.sign_bitmaps(bitmaps[k][c]) for c, sl in enumerate(sls)], ctx)
            for c in range(C):
                g, w = got[k * C + c], want[c]
                assert np.array_equal(g.data, w.data), (cfg, k, c, len(g.data), len(w.data))
                assert np.array_equal(g.address, w.address), (cfg, k, c)
                mine = slicers()[c]
                run.state_into(k * C + c, mine)
                assert bytes(mine._state) == bytes(sls[c]._state), (cfg, k, c)
                produced += len(w.data)
        assert produced > 0
        # pieces of the rows come back the same as the whole
        some = run.fetch(C, 2 * C)
        for j in range(2 * C):
            assert np.array_equal(some[j].data, got[C + j].data) and np.array_equal(some[j].address, got[C + j].address)
    finally:
        eng.close()


def pk(pkts):
    return (np.array([p.streamaddress for p in pkts], dtype=np.int64), np.array([len(p.data) for p in pkts], dtype=np.int64),
            np.array([p.BytesCorrected for p in pkts], dtype=np.int64), np.array([b for p in pkts for b in p.data], dtype=np.uint8))


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("mode,cfg", [("qpsk2400_il2p", "qpsk_2400.json"), ("bpsk300_il2p", "bpsk_300.json")])
def test_recordings_executor_matches_the_oracle_on_packet_bearing_audio(mode, cfg, fused, monkeypatch):
    """process_recordings_device on four different packet-bearing recordings: slicer bytes, addresses and packets of every chain of
    every recording equal the oracle's (which the reference's goldens pin)."""
    from pymodem_amd import chain_builder as cb, siggen
    from pymodem_amd.loop_batch import process_recordings_device
    rate = 48000
    lines = lines_of(cfg)
    recs = []
    for k in range(4):
        audio, _ = siggen.recording(mode, rate, packets=2, seed=50 + k, noise_sigma=900.0 + 300.0 * k, payload_len=(20, 40))
        recs.append(audio)
    n = min(len(r) for r in recs)
    recs = [r[:n] for r in recs]
    chain_sets = [[cb.build_chain(rate, line) for line in lines] for _ in recs]
    stages = {}
    monkeypatch.setenv("PYMODEM_AMD_LOOP_FUSED_SLICERS", str(fused))      # the slicers inside the engine / pm_slice_batch afterwards
    got = process_recordings_device(chain_sets, recs, chunk=8192, stages=stages)
    assert bool(stages.get("fused_slicers")) == bool(fused)
    decoded = 0
    for k, audio in enumerate(recs):
        for c, line in enumerate(lines):
            want = O.run_chain(O.build_chain(rate, line), audio, canon=True)
            sl = stages["sliced"][k][c]
            assert np.array_equal(sl.data, want["slice_data"]) and np.array_equal(sl.address, want["slice_addr"]), (cfg, k, c)
            for a, b in zip(pk(got[k][c]), pk(want["packets"])):
                assert np.array_equal(a, b), (cfg, k, c)
            decoded += len(want["packets"])
    assert decoded > 0


def test_recordings_executor_deferred_batches_and_the_way_back(monkeypatch):
    """defer=True: the call returns when the GPU is through and hands the host's share back as a function -- two batches with their
    output rows in two slots, the first one's share run after the second batch's engine run, equal what the plain call returns.  And
    the way back: when a stream's output outgrows its room (forced here: room for eight bytes) the run is sliced from its sign bitmaps
    instead, with the same result."""
    from pymodem_amd import chain_builder as cb, siggen
    from pymodem_amd import loop_batch as lbm
    from pymodem_amd.loop_batch import process_recordings_device
    rate = 48000
    lines = lines_of("bpsk_300.json")
    recs = []
    for k in range(4):
        audio, _ = siggen.recording("bpsk300_il2p", rate, packets=2, seed=80 + k, noise_sigma=1000.0, payload_len=(20, 40))
        recs.append(audio)
    n = min(len(r) for r in recs)
    recs = [r[:n] for r in recs]

    def chains(count):
        return [[cb.build_chain(rate, line) for line in lines] for _ in range(count)]
    want = process_recordings_device(chains(4), recs, chunk=8192)
    assert sum(len(p) for rec in want for p in rec) > 0
    first = process_recordings_device(chains(2), recs[:2], chunk=8192, slot=0, defer=True)
    second = process_recordings_device(chains(2), recs[2:], chunk=8192, slot=1, defer=True)
    got = first() + second()
    for k in range(4):
        for c in range(len(lines)):
            for a, b in zip(pk(got[k][c]), pk(want[k][c])):
                assert np.array_equal(a, b), (k, c)
    # no room: every row flags it, the executor goes the other way
    real = lbm.LoopBatch._sliced_room

    def tiny(self, r, nout, slicers, slot):
        cap, rows, data, steps, recs_ = real(self, r, nout, slicers, slot)
        return 8, rows, data, steps, recs_
    monkeypatch.setattr(lbm.LoopBatch, "_sliced_room", tiny)
    stages = {}
    again = process_recordings_device(chains(4), recs, chunk=8192, stages=stages)
    assert not stages.get("fused_slicers")
    for k in range(4):
        for c in range(len(lines)):
            for a, b in zip(pk(again[k][c]), pk(want[k][c])):
                assert np.array_equal(a, b), (k, c)


@pytest.mark.parametrize("cfg,rate", [("bpsk_300.json", 48000), ("qpsk_2400.json", 48000)])
def test_sliced_runs_on_short_and_unaligned_recordings(cfg, rate):
    """The edges of a sliced run: recordings barely longer than the chain's filters (one output, a few outputs, one word and a bit), and
    recordings that do not start on a 16-byte boundary (the engine then finds the AGC's `normal` the reference's way, chunk by chunk):
    bytes, addresses and end states equal run() + slice_batch."""
    import pymodem_amd
    from pymodem_amd import chain_builder as cb
    from pymodem_amd.loop_batch import LoopBatch
    from pymodem_amd.slicer import slice_batch
    ctx = pymodem_amd.Context.default()
    group = group_modems(cfg, rate, [1500.0, 1510.0])
    modems = [m for _, m in group]

    def slicers():
        return [cb.build_chain(rate, line)[2] for line, _ in group]
    eng = LoopBatch(modems, recordings=3, ctx=ctx, chunk=2048)
    try:
        shortest = 1 - eng.geometry(0)[0]                     # samples that give one output
        for n, shift in [(shortest, 0), (shortest + 5, 0), (shortest + 70, 0), (shortest + 2048 + 3, 0), (9000 + shortest, 3), (9000 + shortest, 5)]:
            base = [ctx.upload(noise_i16(n + 8, seed=900 + k + n % 97, sigma=4000.0)) for k in range(3)]
            dev = [b.view(shift, n) for b in base]            # shift != 0: int16 pointers off the 16-byte grid
            run = eng.run_sliced(dev, slicers())
            ctx.sync()
            assert run.ok()
            got = run.fetch(0, run.rows)
            bitmaps = eng.run(dev)
            ctx.sync()
            assert run.nout == bitmaps[0][0].n and run.nout == n - shortest + 1
            C = len(group)
            for k in range(3):
                sls = slicers()
                want = slice_batch(sls, [sl.sign_bitmaps(bitmaps[k][c]) for c, sl in enumerate(sls)], ctx)
                for c in range(C):
                    g, w = got[k * C + c], want[c]
                    assert np.array_equal(g.data, w.data) and np.array_equal(g.address, w.address), (cfg, n, shift, k, c)
                    mine = slicers()[c]
                    run.state_into(k * C + c, mine)
                    assert bytes(mine._state) == bytes(sls[c]._state), (cfg, n, shift, k, c)
    finally:
        eng.close()


def test_sliced_run_of_the_qpsk_modem():
    """QPSKModem (psk.py:197-476: one input, two low-passed arms) through a sliced run with a quadrature slicer: bytes, addresses and
    end states equal run() + slice_batch."""
    import pymodem_amd
    from pymodem_amd import chain_builder as cb
    from pymodem_amd.loop_batch import LoopBatch
    from pymodem_amd.slicer import slice_batch
    ctx = pymodem_amd.Context.default()
    spec = {"type": "qpsk", "config": "2400", "options": {}}
    modems = [cb.ModemConfigurator(48000, spec)]

    def slicers():
        return [cb.SlicerConfigurator(48000, {"type": "quadrature", "config": "qpsk_2400", "options": {}})]
    recs = [noise_i16(20011, seed=17 + k, sigma=3000.0) for k in range(3)]
    eng = LoopBatch(modems, recordings=3, ctx=ctx, chunk=4096)
    try:
        dev = [ctx.upload(r) for r in recs]
        run = eng.run_sliced(dev, slicers())
        ctx.sync()
        assert run.ok()
        got = run.fetch(0, run.rows)
        bitmaps = eng.run(dev)
        ctx.sync()
        for k in range(3):
            sls = slicers()
            want = slice_batch(sls, [sls[0].sign_bitmaps(bitmaps[k][0])], ctx)[0]
            assert len(want.data) > 100
            assert np.array_equal(got[k].data, want.data) and np.array_equal(got[k].address, want.address), k
            mine = slicers()[0]
            run.state_into(k, mine)
            assert bytes(mine._state) == bytes(sls[0]._state), k
    finally:
        eng.close()
