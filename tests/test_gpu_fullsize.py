"""BASELINE.json's full size (10 min @ 48 kHz = 28.8 M samples) on the GPU, checked through size-independent properties:
FIR-class kernels on random output windows against the oracle's exact dot products (same fma order -> bit-exact), the sign
bitmap against a direct comparison, and the chunk-parallel slicer against the oracle's sequential C slicer over the WHOLE
stream (bytes and addresses identical)."""
import ctypes

import numpy as np
import pytest

from conftest import noise_i16
from oracle import oracle as O

pytestmark = pytest.mark.gpu
N = 28_800_000


@pytest.fixture(scope="module")
def ctx():
    import pymodem_amd
    return pymodem_amd.Context.default()


def windows(rng, nout, m, count=300):
    ks = np.concatenate([[0, 1, 2047, 2048, 2049, nout - 1, nout - 2], rng.integers(0, nout, count)])
    return np.unique(ks)


def test_afsk_modem_stages_at_full_size(ctx, config_lines):
    from pymodem_amd import chain_builder as cb
    from pymodem_amd._native import check, lib
    rng = np.random.default_rng(7)
    audio = noise_i16(N)
    modem = cb.ModemConfigurator(48000, config_lines("afsk_1200_ax25_super_opt.json")[1]["modem"])
    d_audio = ctx.upload(audio)
    bpf = modem.front_end(d_audio)
    h_bpf = bpf.download()
    m = len(modem.input_bpf)
    assert len(h_bpf) == N - m + 1
    for k in windows(rng, len(h_bpf), m):
        assert h_bpf[k] == O.fir_canon(audio[k:k + m], modem.input_bpf)[0]
    # correlators on the GPU's own band-passed stream
    mc = len(modem.mark_correlator_i)
    corr = ctx.empty(len(h_bpf) - mc + 1, np.float64)
    t = [ctx.upload(v) for v in (modem.mark_correlator_i, modem.mark_correlator_q, modem.space_correlator_i, modem.space_correlator_q)]
    check(lib().pm_afsk_correlate(ctx.handle, bpf.ptr, bpf.n, t[0].ptr, t[1].ptr, t[2].ptr, t[3].ptr, mc, corr.ptr))
    h_corr = corr.download()
    for k in windows(rng, len(h_corr), mc):
        want = O.afsk_correlate_canon(h_bpf[k:k + mc], modem.mark_correlator_i, modem.mark_correlator_q,
                                      modem.space_correlator_i, modem.space_correlator_q)[0]
        assert h_corr[k] == want
    # output low-pass, float64 out and fused sign bitmap
    ml = len(modem.output_lpf)
    taps = ctx.upload(modem.output_lpf)
    lpf = ctx.empty(len(h_corr) - ml + 1, np.float64)
    check(lib().pm_fir_valid_f64(ctx.handle, corr.ptr, corr.n, taps.ptr, ml, lpf.ptr, 0))
    h_lpf = lpf.download()
    for k in windows(rng, len(h_lpf), ml):
        assert h_lpf[k] == O.fir_canon(h_corr[k:k + ml], modem.output_lpf)[0]
    bits = ctx.empty((len(h_lpf) + 63) // 64 + 1, np.uint64)
    check(lib().pm_fir_signs_f64(ctx.handle, corr.ptr, corr.n, taps.ptr, ml, bits.ptr, 0))
    got = np.unpackbits(bits.download().view(np.uint8), bitorder="little")[:len(h_lpf)].astype(bool)
    assert np.array_equal(got, h_lpf >= 0)
    # whole-stream slicer: chunk-parallel fixed point vs the sequential oracle
    from pymodem_amd.slicer import BinarySlicer
    s = BinarySlicer(sample_rate=48000.0, config="1200")
    s.StringOptionsRetune({"lock_rate": "0.77"})
    out = s.slice(lpf)
    d, a = O.BinarySlicer(48000.0, "1200", {"lock_rate": "0.77"}).slice(h_lpf)
    assert np.array_equal(out.data, d) and np.array_equal(out.address, a)
    assert len(d) > 80000 and s.last_stats["iterations"] < 200


def test_quadrature_slicer_and_agc_at_full_size(ctx):
    """QPSK-2400 slicer (lock 0.98, differential demap across chunk boundaries) and the chunk-parallel AGC over 28.8 M samples."""
    from pymodem_amd._native import AGCParams, check, lib
    from pymodem_amd.data_classes import DeviceIQ
    from pymodem_amd.slicer import QuadratureSlicer
    rng = np.random.default_rng(11)
    xi = np.convolve(rng.standard_normal(N + 39), np.hanning(40), "valid")
    xq = np.convolve(rng.standard_normal(N + 39), np.hanning(40), "valid")
    s = QuadratureSlicer(sample_rate=48000.0, config="qpsk_2400")
    s.StringOptionsRetune({"lock_rate": "0.98"})
    out = s.slice(DeviceIQ(ctx.upload(xi), ctx.upload(xq)))
    d, a = O.QuadratureSlicer(48000.0, "qpsk_2400", {"lock_rate": "0.98"}).slice((xi, xq))
    assert np.array_equal(out.data, d) and np.array_equal(out.address, a)
    # AGC: bursty envelope so that attack, sustain and decay all occur many times
    env = 0.2 + np.abs(np.sin(np.arange(N) / 300000.0))
    buf = xi * env * 100.0
    want = buf.copy()
    st_o = np.zeros(2)
    O.agc_apply(want, 48000.0, 500.0, 1.0, 50.0, 1.0, state=st_o)
    dbuf = ctx.upload(buf)
    st = (ctypes.c_double * 2)(0.0, 0.0)
    p = AGCParams(500.0, 50.0, 1.0, 48000.0, 1.0)
    check(lib().pm_agc_apply(ctx.handle, dbuf.ptr, N, ctypes.byref(p), st))
    assert np.array_equal(dbuf.download(), want)
    assert st[0] == st_o[0] and st[1] == st_o[1]


def test_headline_workload_end_to_end_at_full_size(ctx):
    """The bench's headline workload (8 AFSK-1200 chains, 28.8 M-sample packet-bearing buffer) through the group executor; three of
    the chains (the one with its own correlators, the first and the last of the shared-mark group) against the oracle: slicer
    bytes, addresses and every packet identical.  (tests/fullsize_parity.py does all chains of all four workloads.)"""
    import bench
    from pymodem_amd import chain_builder as cb, chain_execute as ce

    class A:
        pass
    args = A()
    args.samples, args.rate, args.workload, args.buffer = N, 48000, "afsk_1200_super_opt", "signal"
    audio = bench.make_buffer(args)
    factory, cpg, _ = bench.WORKLOADS[args.workload]
    lines = [factory(c) for c in range(cpg)]
    from conftest import oracle_chains
    wanted = oracle_chains([(lines[c], audio) for c in (0, 1, 7)])
    stages = {}
    pk = ce.process_chains_device([cb.build_chain(48000, l) for l in lines], audio, stages=stages)
    assert [len(p) for p in pk] == [691, 690, 690, 610, 500, 420, 340, 260]
    for k, c in enumerate((0, 1, 7)):
        r = wanted[k].result()
        sl = stages["sliced"][c]
        assert np.array_equal(sl.data, r["slice_data"]) and np.array_equal(sl.address, r["slice_addr"]), c
        got = [(p.streamaddress, bytes(bytearray(p.data))) for p in pk[c]]
        want = [(p.streamaddress, bytes(bytearray(p.data))) for p in r["packets"]]
        assert got == want, c


def test_native_executor_at_full_size_equals_the_group_executor(ctx):
    """The same recording through pm_pipe_* (band-pass and low-pass sums as int8 digit products on the matrix pipe, exact recomputation
    of the uncertain decisions from the audio): every packet row of every chain and the de-dup equal to the group executor's, which
    the test above pins to the oracle.  Two recordings in flight, the second one the buffer shifted by 7 samples (unaligned audio
    takes the binary64 band-pass)."""
    import bench
    from pymodem_amd import chain_builder as cb, chain_execute as ce
    from pymodem_amd.packet_meta import PacketTable

    class A:
        pass
    args = A()
    args.samples, args.rate, args.workload, args.buffer = N, 48000, "afsk_1200_super_opt", "signal"
    audio = bench.make_buffer(args)
    factory, cpg, _ = bench.WORKLOADS[args.workload]
    lines = [factory(c) for c in range(cpg)]
    names = [l["object_name"] for l in lines]
    d0 = ctx.upload(audio)
    shifted = d0.view(7, len(audio) - 7)
    ctx.sync()
    pipe = ce.NativePipeline([cb.build_chain(48000, l) for l in lines], len(audio), 48000 / 40, ctx=ctx)
    tickets = [pipe.submit(d0), pipe.submit(shifted)]
    for t, a in zip(tickets, [audio, audio[7:]]):
        rows = ce.process_chains_table([cb.build_chain(48000, l) for l in lines], a)
        want = PacketTable(dict(rows), names).correlate(48000 / 40)
        table = pipe.table(t)
        assert table.counts == want.counts
        at = 0
        for c in range(len(lines)):
            got = table.rows[at:at + table.counts[c]]
            at += table.counts[c]
            assert all(np.array_equal(got[f], rows[c][f]) for f in got.dtype.names if f != "correlated_count"), c
        assert np.array_equal(table.unique_idx, want.unique_idx) and table.unique_decoders == want.unique_decoders
    assert table.counts[0] > 600
    pipe.close()
