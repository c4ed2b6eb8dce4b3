"""Rank 0's share of a step in an 8-GPU run, timed on the CPU (no GPU, no process group): what `bench.py`'s `post()` does per recording
once the one all_gather has delivered every rank's wire stream -- index 64 chains' records (`dist.table_from_exchange`,
pm_packets_index) and de-duplicate them in config order (`PacketTable.correlate`, pm_correlate: packet_meta.py:230-271 behind
pymodem.py:170-175) -- at the headline workload's packet density: 725 packets per ten-minute recording, each seen by ~95 % of the 64 chains
(the space_gain sweep continued in finer steps, as `--gpus 8` runs it), 44 000 records and 4.8 MB of wire data per recording.

The budget: the headline's steady step is 0.65 ms per recording (profiles/r05_*, BENCH_r04.json) and `post()` keeps FOUR recordings in
flight on four threads, both calls native with the interpreter lock released -- so rank 0 keeps up as long as ONE recording costs less
than 4 x 0.65 = 2.6 ms of one thread.  Measured: 1.85 ms on the build container's 2.1 GHz Xeon (29 % margin; a shared machine whose
threads do not scale), 0.80 ms on the GPU boxes' EPYC 9575F, where the four-thread stage gets through a recording every 0.46 ms -- 29 %
inside the step.  A scaling curve has never been measured (no 8-GPU node in any round): this pins the one cost that grows with the
world size -- everything else a rank does is what it does at N = 1."""
import ctypes
import time

import numpy as np

WORLD, CHAINS, PACKETS = 8, 64, 725
STEP_MS, POST_THREADS = 0.65, 4


def _recording(seed=5):
    """-> (wire streams of the 8 ranks, records per chain): what rank 0 holds after the exchange of one recording."""
    from pymodem_amd import siggen
    from pymodem_amd._native import check, lib, packet_dtype
    from pymodem_amd.packet_meta import _stamp
    rng = np.random.default_rng(seed)
    base = np.zeros(PACKETS, dtype=packet_dtype())
    addr = 20000
    for k in range(PACKETS):
        frame = siggen.ax25_ui_frame("CQ", f"N0CAL{k % 10}", [int(c) for c in rng.integers(32, 127, int(rng.integers(20, 81)))], src_ssid=k % 16)
        crc = lib().pm_crc16_ccitt((ctypes.c_uint8 * len(frame))(*frame), len(frame))
        data = list(frame) + [crc & 255, crc >> 8]
        base[k]["data"][:len(data)] = data
        base[k]["len"] = len(data)
        base[k]["calculated_crc"] = base[k]["carried_crc"] = crc
        base[k]["valid_crc"] = base[k]["valid_header"] = 1
        addr += int(rng.integers(30000, 50000))
        base[k]["streamaddress"] = addr
    per = CHAINS // WORLD
    streams, counts = [], np.zeros(CHAINS, dtype=np.int64)
    for r in range(WORLD):
        parts = []
        for c in range(r * per, (r + 1) * per):
            rows = base[rng.random(PACKETS) < 0.95].copy()
            rows["streamaddress"] += rng.integers(-3, 4, len(rows))          # the chains' clocks put a packet's last byte a sample or two apart
            bad = rng.random(len(rows)) < 0.004                               # (the headline: 21 rejected frames beside 690 packets x 8 chains)
            rows["valid_crc"][bad] = 0
            rows["calculated_crc"][bad] ^= 0x5A5A
            _stamp(rows, c)
            counts[c] = len(rows)
            parts.append(rows)
        mine = np.ascontiguousarray(np.concatenate(parts))
        need = check(lib().pm_packets_pack(mine.ctypes.data_as(ctypes.c_void_p), len(mine), None, 0))
        buf = np.zeros(need, dtype=np.uint8)
        check(lib().pm_packets_pack(mine.ctypes.data_as(ctypes.c_void_p), len(mine), buf.ctypes.data_as(ctypes.c_void_p), need))
        streams.append(buf)
    return streams, counts.tolist()


def test_rank_0_keeps_up_with_the_step_at_world_8():
    from concurrent.futures import ThreadPoolExecutor
    from pymodem_amd import dist as pdist
    streams, counts = _recording()
    names = [f"chain {c}" for c in range(CHAINS)]
    assert sum(counts) > 40000 and sum(len(s) for s in streams) > 4_000_000

    def dedupe(_):
        table = pdist.table_from_exchange(("streams", list(streams), counts), names)
        return table.correlate(48000 / 40)

    table = dedupe(0)
    # every packet is found once, credited to (nearly) every chain that carried it, in config order
    assert table.CountGood() == PACKETS
    seen = [len(d) for d in table.unique_decoders]
    assert max(seen) <= CHAINS and sum(seen) == int(np.count_nonzero(table.heads["valid_crc"]))
    assert all(d == sorted(d, key=names.index) for d in table.unique_decoders[:50])
    for _ in range(4):
        dedupe(0)
    best = 1e9
    for _ in range(3):                                        # the best of three: the build container is a shared machine
        t0 = time.perf_counter()
        for k in range(16):
            dedupe(k)
        best = min(best, (time.perf_counter() - t0) / 16 * 1e3)
    budget = STEP_MS * POST_THREADS
    print(f"rank 0 at world {WORLD}: {best:.2f} ms of one thread per recording, budget {budget:.2f} ms ({POST_THREADS} post threads x {STEP_MS} ms step)")
    assert best < budget, (best, budget)
    # ... and the four-thread stage itself gets through the recordings (a throughput figure for the log; no assertion: thread scaling is the host's)
    with ThreadPoolExecutor(POST_THREADS) as pool:
        t0 = time.perf_counter()
        list(pool.map(dedupe, range(32)))
        print(f"through {POST_THREADS} threads: {(time.perf_counter() - t0) / 32 * 1e3:.2f} ms per recording")
