"""Native host-integer stages (pm_lfsr_unscramble, pm_codec_*, pm_correlate; C++ in pymodem_amd/csrc/pm_codec.cpp)
against the oracle's plain-Python restatement on adversarial random streams, and against the reference goldens.
No GPU needed: these entry points never touch HIP."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle as O


def biased_bytes(rng, n, p_one):
    bits = (rng.random(n * 8) < p_one).astype(np.uint8)
    return np.packbits(bits)


def pk(pkts):
    return [(int(p.streamaddress), [int(b) for b in p.data], int(p.BytesCorrected)) for p in pkts]


@pytest.mark.parametrize("poly,inv", [(0x1, False), (0x1, True), (0x3, True), (0x3, False), (0x63003, True), (0x211, False), (0x10801, False)])
def test_lfsr_matches_oracle(poly, inv):
    from pymodem_amd.data_classes import AddressedArray
    from pymodem_amd.lfsr import LFSR
    rng = np.random.default_rng(poly)
    for n in [0, 1, 2, 7, 8, 9, 63, 64, 65, 1000, 70001]:
        data = rng.integers(0, 256, n, dtype=np.uint8)
        addr = np.arange(n, dtype=np.int64) * 3 + 5
        st = LFSR(poly=poly, invert=inv)
        o = O.LFSR(poly, inv)
        # two calls: the shift register carries over between them
        cut = n // 3
        got = np.concatenate([st.stream_unscramble_8bit(AddressedArray(data[:cut], addr[:cut])).data,
                              st.stream_unscramble_8bit(AddressedArray(data[cut:], addr[cut:])).data])
        want = np.concatenate([o.stream_unscramble_8bit(data[:cut]), o.stream_unscramble_8bit(data[cut:])])
        assert np.array_equal(got, want), (poly, n)
        assert st.shift_register == o.sr.value


@pytest.mark.parametrize("p_one", [0.5, 0.7, 0.85, 0.93, 0.3])
def test_ax25_matches_oracle_on_random_streams(p_one):
    """Dense ones exercise flags, stuffed zeros, aborts and >1023-byte frames (ax25.py:36-50,70-88)."""
    from pymodem_amd.codecs import AX25Codec
    from pymodem_amd.data_classes import AddressedArray
    rng = np.random.default_rng(int(p_one * 100))
    for n in [1, 50, 4000, 60000]:
        data = biased_bytes(rng, n, p_one)
        addr = np.cumsum(rng.integers(1, 50, n)).astype(np.int64)
        cut = n // 2
        c = AX25Codec(ident="x")
        got = c.decode(AddressedArray(data[:cut], addr[:cut])) + c.decode(AddressedArray(data[cut:], addr[cut:]))
        o = O.AX25Codec(ident="x")
        want = o.decode(data[:cut], addr[:cut]) + o.decode(data[cut:], addr[cut:])
        assert pk(got) == pk(want), (p_one, n, len(got), len(want))
    # frames are actually found in the unbiased regime (dense ones mostly produce aborts)
    assert p_one != 0.5 or len(want) > 0


@pytest.mark.parametrize("p_one", [0.5, 0.6, 0.75, 0.4])
def test_ax25_in_pieces_of_any_size_matches_oracle(p_one):
    """The decoder's registers carry over between calls (ax25.py:17-23) and the native decoder skims whole calls 64 bits at a time for the
    flags that can close a frame: pieces of 1 ... 3000 bytes, cut anywhere -- inside flags, runs of ones, stuffed zeros -- with real frames
    planted between the random bits, must give the reference's packets, addresses and order."""
    from pymodem_amd import siggen
    from pymodem_amd.codecs import AX25Codec
    from pymodem_amd.data_classes import AddressedArray
    rng = np.random.default_rng(int(p_one * 1000))
    bits = []
    for k in range(60):
        frame = siggen.ax25_ui_frame("CQ", f"N0CAL{k % 10}", [int(c) for c in rng.integers(32, 127, int(rng.integers(1, 120)))])
        fb = np.array(siggen.ax25_hdlc_bits(frame), dtype=np.uint8)
        if k % 7 == 3:
            fb[int(rng.integers(20, len(fb) - 20))] ^= 1                      # a damaged frame: closes (or not) wherever the bits say
        if k % 11 == 5:
            fb = np.concatenate([fb[:len(fb) // 2], np.ones(9, dtype=np.uint8), fb[len(fb) // 2:]])     # an abort in mid-frame: the bytes stay
        bits.append(fb)
        bits.append((rng.random(int(rng.integers(0, 6000))) < p_one).astype(np.uint8))
    b = np.concatenate(bits)
    data = np.packbits(b[:len(b) // 8 * 8])
    n = len(data)
    addr = np.cumsum(rng.integers(1, 50, n)).astype(np.int64)
    want = pk(O.AX25Codec(ident="x").decode(data, addr))
    assert len(want) >= 40
    for trial in range(4):
        c = AX25Codec(ident="x")
        got, at = [], 0
        top = [3000, 200, 30, 9][trial]
        while at < n:
            step = int(rng.integers(1, top))
            got += c.decode(AddressedArray(data[at:at + step], addr[at:at + step]))
            at += step
        assert pk(got) == want, (p_one, trial, len(got), len(want))


def test_ax25_frames_longer_than_a_row():
    """A flag after a long stretch without one closes a frame of any length in the reference (the byte counter wraps at 1023, the bytes
    stay: ax25.py:41-47).  The native row keeps the first PM_PKT_MAX bytes; address, count, CRC fields and validity are the whole
    frame's."""
    import ctypes
    from pymodem_amd._native import PKT_MAX, Packet, check, lib
    rng = np.random.default_rng(5)
    long_frames = 0
    for trial in range(12):
        n = 20000
        bits = (rng.random(n * 8) < rng.choice([0.3, 0.2, 0.1])).astype(np.uint8)
        for pos in rng.integers(0, n * 8 - 8, 60):
            bits[pos:pos + 8] = [0, 1, 1, 1, 1, 1, 1, 0]
        data = np.packbits(bits)
        addr = (np.arange(n, dtype=np.int64) + 1) * 40
        want = O.AX25Codec(ident="x").decode(data, addr)
        h = ctypes.c_void_p()
        check(lib().pm_codec_create(0, 1, 0, 0, 2, 0, ctypes.byref(h)))
        pend, cnt = ctypes.c_int64(), ctypes.c_int64()
        check(lib().pm_codec_decode(h, data.ctypes.data_as(ctypes.c_void_p), addr.ctypes.data_as(ctypes.c_void_p), n, ctypes.byref(pend)))
        rows = (Packet * max(pend.value, 1))()
        check(lib().pm_codec_fetch(h, rows, pend.value, ctypes.byref(cnt)))
        lib().pm_codec_destroy(h)
        assert cnt.value == len(want)
        for r, w in zip(rows, want):
            full = bytes(bytearray(w.data))
            long_frames += len(full) > PKT_MAX
            w.check()
            assert r.streamaddress == w.streamaddress and r.len == min(len(full), PKT_MAX) and bytes(r.data[:r.len]) == full[:PKT_MAX]
            assert (r.calculated_crc, r.carried_crc, bool(r.valid_crc)) == (w.CalculatedCRC, w.CarriedCRC, bool(w.ValidCRC))
    assert long_frames >= 3


def il2p_stream(rng, n, sync_every):
    """Random bytes with the IL2P sync word 0xF15E48 planted (sometimes with bit errors) so headers get attempted."""
    data = rng.integers(0, 256, n, dtype=np.uint8)
    for pos in range(10, n - 4, sync_every):
        w = [0xF1, 0x5E, 0x48]
        if rng.random() < 0.5:
            w[rng.integers(0, 3)] ^= 1 << int(rng.integers(0, 8))
        data[pos:pos + 3] = w
    return data


@pytest.mark.parametrize("tol,crc,md", [(0, True, 0), (2, True, 0), (2, False, 1), (1, True, 0)])
def test_il2p_matches_oracle_on_random_streams(tol, crc, md):
    from pymodem_amd.codecs import IL2PCodec
    from pymodem_amd.data_classes import AddressedArray
    rng = np.random.default_rng(tol * 10 + md)
    for n, every in [(3000, 40), (40000, 300), (40000, 2000)]:
        data = il2p_stream(rng, n, every)
        addr = np.arange(n, dtype=np.int64) * 40 + 7
        c = IL2PCodec(ident="x", crc=crc, min_dist=md, sync_tol=tol)
        o = O.IL2PCodec("x", crc, False, md, tol)
        cut = n // 3
        got = c.decode(AddressedArray(data[:cut], addr[:cut])) + c.decode(AddressedArray(data[cut:], addr[cut:]))
        want = o.decode(data[:cut], addr[:cut]) + o.decode(data[cut:], addr[cut:])
        assert pk(got) == pk(want), (tol, n, len(got), len(want))


def test_codecs_reproduce_reference_packets_on_bundled_recording(golden, config_lines):
    """Golden slicer bytes of the bundled recording -> native LFSR + codec + Correlate == the reference's packets."""
    from pymodem_amd import chain_builder as cb, packet_meta as pm
    from pymodem_amd.data_classes import AddressedArray
    g = golden("wav_chains")
    summ = json.load(open(os.path.join(GOLDEN, "wav_chains_summary.json")))
    for cfg in ["afsk_300.json", "afsk_300_pll.json", "afsk_300_ax25.json"]:
        k = cfg[:-5]
        res = pm.PacketMetaArray()
        for ci, line in enumerate(config_lines(cfg)):
            prefix = f"{k}__c{ci}"
            stream = cb.StreamConfigurator(line["stream"])
            codec = cb.CodecConfigurator(line["codec"], line["object_name"])
            lf = stream.stream_unscramble_8bit(AddressedArray(g[prefix + "_slice_data"], g[prefix + "_slice_addr"]))
            assert np.array_equal(lf.data, g[prefix + "_lfsr_data"])
            pkts = codec.decode(lf)
            assert np.array_equal(np.array([p.streamaddress for p in pkts], dtype=np.int64), g[prefix + "_pkt_addr"])
            assert np.array_equal(np.array([b for p in pkts for b in p.data], dtype=np.uint8), g[prefix + "_pkt_data"])
            assert np.array_equal(np.array([p.BytesCorrected for p in pkts], dtype=np.int64), g[prefix + "_pkt_corrected"])
            res.add(pkts)
        res.CalcCRCs()
        res.Correlate(address_distance=8000 / 40)
        assert res.CountGood() == summ[k]["good"] and res.CountBad() == summ[k]["bad"]
        u = res.unique_packet_array
        assert np.array_equal(np.array([p.streamaddress for p in u], dtype=np.int64), g[k + "__uniq_addr"])
        assert [list(p.CorrelatedDecoders) for p in u] == summ[k]["uniq_decoders"]
        assert dict(res.DecoderHistogram) == summ[k]["hist"] and dict(res.DecoderUniqueHistogram) == summ[k]["uniq_hist"]


def test_packet_table_path_matches_the_object_path(golden, config_lines):
    """decode_rows -> PacketTable.correlate (all native, no PacketMeta objects) == decode -> CalcCRCs -> Correlate."""
    from pymodem_amd import chain_builder as cb
    from pymodem_amd.data_classes import AddressedArray
    from pymodem_amd.packet_meta import PacketTable
    g = golden("wav_chains")
    summ = json.load(open(os.path.join(GOLDEN, "wav_chains_summary.json")))
    for cfg in ["afsk_300.json", "afsk_300_ax25.json"]:
        k = cfg[:-5]
        lines = config_lines(cfg)
        rows = {}
        for ci, line in enumerate(lines):
            lf = cb.StreamConfigurator(line["stream"]).stream_unscramble_8bit(
                AddressedArray(g[f"{k}__c{ci}_slice_data"], g[f"{k}__c{ci}_slice_addr"]))
            rows[ci] = cb.CodecConfigurator(line["codec"], line["object_name"]).decode_rows(lf)
        t = PacketTable(rows, [l["object_name"] for l in lines]).correlate(8000 / 40)
        assert t.CountGood() == summ[k]["good"] and t.CountBad() == summ[k]["bad"]
        assert np.array_equal(t.rows["streamaddress"][t.unique_idx], g[k + "__uniq_addr"])
        assert np.array_equal(t.rows["calculated_crc"][t.unique_idx], g[k + "__uniq_crc"])
        assert t.unique_decoders == summ[k]["uniq_decoders"]
        flat = np.stack([t.rows["valid_crc"], t.rows["valid_header"], t.rows["calculated_crc"], t.rows["carried_crc"]], axis=1)
        assert np.array_equal(flat.astype(np.int64), g[k + "__raw_valid"])
        u = t.unique_packets()
        assert [p.streamaddress for p in u] == g[k + "__uniq_addr"].tolist() and all(p.ValidCRC for p in u)
        for ci in range(len(lines)):
            assert [p.streamaddress for p in t.packets(ci)] == g[f"{k}__c{ci}_pkt_addr"].tolist()


@pytest.mark.parametrize("tol", [0, 1, 2, 3, 5])
def test_il2p_sync_search_at_every_bit_offset(tol):
    """Real IL2P frames (pymodem_amd.siggen) dropped at random BIT offsets into random bits, their sync word 0xF15E48 hit by up to
    tol+1 bit errors anywhere in it: frames within the tolerance must come out, the others must not, exactly as the bit-serial
    oracle decides -- the table-filtered byte-wise search sees every offset."""
    from pymodem_amd import siggen
    from pymodem_amd.codecs import IL2PCodec
    from pymodem_amd.data_classes import AddressedArray
    rng = np.random.default_rng(77 + tol)
    chunks, planted, within = [], 0, 0
    for k in range(160):
        chunks.append(rng.integers(0, 2, int(rng.integers(40, 400)), dtype=np.uint8))
        info = [int(c) for c in rng.integers(32, 127, int(rng.integers(1, 40)))]
        frame = np.array(siggen.il2p_frame_bits("CQ", f"N0CAL{k % 10}", info, src_ssid=k % 16, preamble=2), dtype=np.uint8)
        flips = int(rng.integers(0, tol + 2))
        for f in rng.choice(24, flips, replace=False):
            frame[16 + f] ^= 1                               # the sync word follows the 2 preamble bytes
        chunks.append(frame)
        planted += 1
        within += flips <= tol
    bits = np.concatenate(chunks + [rng.integers(0, 2, 300, dtype=np.uint8)])
    bits = bits[:len(bits) // 8 * 8]
    data = np.packbits(bits)
    addr = np.arange(len(data), dtype=np.int64) * 8 + 3
    c = IL2PCodec(ident="x", crc=True, min_dist=0, sync_tol=tol)
    o = O.IL2PCodec("x", True, False, 0, tol)
    got, want = c.decode(AddressedArray(data, addr)), o.decode(data, addr)
    assert pk(got) == pk(want), (tol, len(got), len(want))
    # the frames inside the tolerance really are decoded (a loose tolerance also fires inside the random filler and eats some)
    assert within > 20 and len(want) >= within * (0.9 if tol <= 2 else 0.4)


@pytest.mark.parametrize("tol", [0, 1, 2, 3])
@pytest.mark.parametrize("pieces", [1, 7])
def test_il2p_sync_word_in_the_tail_of_a_false_header(golden, tol, pieces):
    """Reference golden (tests/golden/il2p_resync.npz, make_goldens.py gen_il2p_resync): valid frames whose sync word starts up to 30
    bits before a false header ends.  Back in the sync search the reference's register holds eight bits and zeros above them
    (il2p.py:146-152), so which of these frames it finds is decided by that register -- the native decoder's skip over infeasible
    input bytes must not look at (or rebuild the register from) input the reference has forgotten.  The oracle is held to the same
    golden.  pieces: the stream fed in one call and in seven (the search state carries over)."""
    from pymodem_amd.codecs import IL2PCodec
    from pymodem_amd.data_classes import AddressedArray
    g = golden("il2p_resync")
    data, addr = g[f"tol{tol}_data"], g[f"tol{tol}_addr"]
    cuts = np.linspace(0, len(data), pieces + 1).astype(int)
    c = IL2PCodec(ident="x", crc=True, min_dist=0, sync_tol=tol)
    o = O.IL2PCodec("x", True, False, 0, tol)
    got, want = [], []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        got += c.decode(AddressedArray(data[lo:hi], addr[lo:hi]))
        want += o.decode(data[lo:hi], addr[lo:hi])
    for name, pkts in (("native", got), ("oracle", want)):
        a = np.array([int(p.streamaddress) for p in pkts], dtype=np.int64)
        dd = np.array([int(b) for p in pkts for b in p.data], dtype=np.uint8)
        cc = np.array([int(p.BytesCorrected) for p in pkts], dtype=np.int64)
        assert np.array_equal(a, g[f"tol{tol}_pkt_addr"]), (name, tol, len(a), int(g[f"tol{tol}_pkt_n"]))
        assert np.array_equal(dd, g[f"tol{tol}_pkt_data"]) and np.array_equal(cc, g[f"tol{tol}_pkt_corrected"]), (name, tol)
    assert int(g[f"tol{tol}_pkt_n"]) >= 60


@pytest.mark.parametrize("threads", [1, 3, 16])
def test_batched_host_stage_equals_the_per_chain_calls(threads, monkeypatch):
    """pm_host_decode_batch + pm_codec_fetch_batch (chain_execute._host_rows) against stream_unscramble_8bit + decode_rows chain by
    chain: same rows, same LFSR registers, codec state carried across two recordings, callers running concurrently."""
    from concurrent.futures import ThreadPoolExecutor
    from pymodem_amd import chain_execute as CE
    from pymodem_amd.codecs import AX25Codec, IL2PCodec
    from pymodem_amd.data_classes import AddressedArray
    from pymodem_amd.lfsr import LFSR
    monkeypatch.setenv("PYMODEM_AMD_HOST_THREADS", str(threads))

    def group(seed):
        rng = np.random.default_rng(seed)
        chains, streams = [], []
        for k in range(9):
            il2p = k % 3 == 2
            codec = IL2PCodec(ident=f"c{k}") if il2p else AX25Codec(ident=f"c{k}")
            chains.append([f"c{k}", None, None, LFSR(poly=[0x3, 0x63003, 0x1][k % 3], invert=k % 2 == 0), codec])
            two = []
            for n in (0 if k == 4 else 20000 + 977 * k, 15000):
                data = il2p_stream(rng, n, 700) if il2p else biased_bytes(rng, n, 0.55)
                two.append(AddressedArray(data, np.cumsum(rng.integers(1, 50, n)).astype(np.int64)))
            streams.append(two)
        return chains, streams

    def run(seed, batched):
        chains, streams = group(seed)
        out = []
        for rec in range(2):
            sliced = [s[rec] for s in streams]
            if batched:
                rows = CE._host_rows(chains, sliced)
            else:
                rows = [ch[4].decode_rows(ch[3].stream_unscramble_8bit(sl)) for ch, sl in zip(chains, sliced)]
            out.append([r.tobytes() for r in rows])
        return out, [ch[3].shift_register for ch in chains]

    want = [run(seed, False) for seed in range(4)]
    assert sum(len(b) for rec in want[0][0] for b in rec) > 0
    with ThreadPoolExecutor(4) as ex:
        got = list(ex.map(lambda seed: run(seed, True), range(4)))
    assert got == want


def test_batched_host_stage_rejects_a_shared_codec():
    import ctypes
    from pymodem_amd._native import HostJob, NativeError, check, lib
    from pymodem_amd.codecs import AX25Codec
    c = AX25Codec(ident="x")
    jobs = (HostJob * 2)()
    for j in range(2):
        jobs[j].codec = c._handle()
    with pytest.raises(NativeError, match="share a codec"):
        check(lib().pm_host_decode_batch(jobs, 2, 4))


def test_recycled_host_blocks_are_never_shared():
    """device._host_block hands a block out again only when nothing refers to it any more (views keep it alive)."""
    from pymodem_amd.device import _host_block
    a = _host_block(3_000_001)
    view = a[:100].view(np.uint8)
    b = _host_block(3_000_001)
    assert b is not a and not np.shares_memory(a, b)
    ida = a.ctypes.data
    del a
    c = _host_block(3_000_001)                      # `view` still refers to the first block
    assert c.ctypes.data != ida
    del view, b, c
    d = _host_block(3_000_001)
    e = _host_block(2_900_000)                      # a second request while d is held: another block
    assert not np.shares_memory(d, e) and d.nbytes >= 3_000_001 and e.nbytes >= 2_900_000


def test_batched_host_stage_survives_a_fork(tmp_path):
    """The library's worker threads do not exist in a forked child: the first batch there starts new ones instead of waiting for
    the parent's (bench.py and the reference's own runner fork)."""
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
from pymodem_amd import chain_execute as CE
from pymodem_amd.codecs import AX25Codec
from pymodem_amd.data_classes import AddressedArray
from pymodem_amd.lfsr import LFSR
def group():
    rng = np.random.default_rng(3)
    chains = [["c%%d" %% k, None, None, LFSR(poly=0x3, invert=True), AX25Codec(ident="c%%d" %% k)] for k in range(6)]
    sliced = [AddressedArray(rng.integers(0, 256, 30000, dtype=np.uint8), np.arange(30000, dtype=np.int64) * 7) for _ in range(6)]
    return chains, sliced
want = [r.tobytes() for r in CE._host_rows(*group())]          # the parent's pool now exists
pid = os.fork()
if pid == 0:
    got = [r.tobytes() for r in CE._host_rows(*group())]
    os._exit(0 if got == want else 3)
_, status = os.waitpid(pid, 0)
sys.exit(os.waitstatus_to_exitcode(status))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], timeout=120, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]


def test_codec_set_source_stamps_the_rows(golden, config_lines):
    """pm_codec_set_source: the index pm_codec_fetch writes into pm_packet.source_decoder (the executor gives every codec its chain's
    place in the config, so that nobody has to walk the rows afterwards)."""
    from pymodem_amd import chain_builder as cb
    from pymodem_amd._native import check, quick
    from pymodem_amd.data_classes import AddressedArray
    g = golden("wav_chains")
    line = config_lines("afsk_300.json")[0]
    for src in (0, 7):
        stream = cb.StreamConfigurator(line["stream"])
        codec = cb.CodecConfigurator(line["codec"], line["object_name"])
        if src:
            check(quick().pm_codec_set_source(codec._handle(), src))
        lf = stream.stream_unscramble_8bit(AddressedArray(g["afsk_300__c0_slice_data"], g["afsk_300__c0_slice_addr"]))
        rows = codec.decode_rows(lf)
        assert len(rows) == len(g["afsk_300__c0_pkt_addr"]) > 3 and (rows["source_decoder"] == src).all()


def test_packet_rows_that_are_views_of_a_byte_block_are_stacked_without_a_copy():
    """The executor inside the library hands a recording's rows out as views of ONE byte array over its own memory: per-chain slices of
    it go back together as one array over the same memory (PacketTable._stack), not through a 7 MB concatenate per recording."""
    from pymodem_amd._native import packet_dtype
    from pymodem_amd.packet_meta import PacketTable
    dt = packet_dtype()
    raw = np.zeros(10 * dt.itemsize + 64, dtype=np.uint8)
    block = raw[64:].view(dt)
    block["streamaddress"] = np.arange(10)
    parts = [block[0:3], block[3:7], block[7:10]]
    whole = PacketTable._stack(parts)
    assert np.shares_memory(whole, raw) and whole["streamaddress"].tolist() == list(range(10))
    whole["len"][4] = 77
    assert block["len"][4] == 77
    gap = PacketTable._stack([block[0:3], block[4:7]])              # not consecutive: a copy
    assert not np.shares_memory(gap, raw) and gap["streamaddress"].tolist() == [0, 1, 2, 4, 5, 6]
