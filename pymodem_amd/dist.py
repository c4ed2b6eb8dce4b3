"""Multi-GPU layer.  demod_chains are independent (they share only the read-only recording, pymodem.py:144-149),
so they are sharded over ranks with NO data-path collective; the one exchange step is the gather of decoded-packet
records to rank 0 for the cross-chain de-dup (PacketMetaArray.Correlate, packet_meta.py:230-271), which in the
reference is a multiprocessing.Queue (pymodem.py:140,157-163).  One process per GPU, torch.distributed:
backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.  Payload is KBs: latency-bound.
"""
import contextlib

import numpy as np

from .packet_meta import PacketMeta, PacketMetaArray

PKT_BYTES = 1280
RECORD = np.dtype([("streamaddress", "<i8"), ("chain", "<i4"), ("len", "<i4"), ("bytes_corrected", "<i4"), ("pad", "<i4"),
                   ("data", "u1", (PKT_BYTES,))])


def shard_chains(nchains, rank, world):
    """Chain c runs on rank c mod world.  Returns this rank's global chain indices, in config order."""
    return [c for c in range(nchains) if c % world == rank]


def pack_packets(packets_by_chain):
    """{global chain index: list[PacketMeta]} -> structured array of fixed-size records."""
    n = sum(len(v) for v in packets_by_chain.values())
    rec = np.zeros(n, dtype=RECORD)
    k = 0
    for chain in sorted(packets_by_chain):
        for p in packets_by_chain[chain]:
            d = p.raw()[:PKT_BYTES]
            rec[k]["streamaddress"], rec[k]["chain"], rec[k]["len"] = p.streamaddress, chain, len(d)
            rec[k]["bytes_corrected"] = p.BytesCorrected
            rec[k]["data"][:len(d)] = np.frombuffer(d, dtype=np.uint8)
            k += 1
    return rec


def unpack_packets(rec, chain_names):
    """records -> {chain index: list[PacketMeta]} (decode order within a chain is preserved)."""
    out = {}
    chains, lens, addrs, corr = rec["chain"].tolist(), rec["len"].tolist(), rec["streamaddress"].tolist(), rec["bytes_corrected"].tolist()
    data = rec["data"]
    for k in range(len(rec)):
        p = PacketMeta.from_bytes(data[k, :lens[k]].tobytes(), addrs[k], chain_names[chains[k]], corr[k])
        out.setdefault(chains[k], []).append(p)
    return out


def gather_packets(packets_by_chain, chain_names, device=None):
    """All ranks call this once per recording.  Rank 0 returns {chain: packets} for ALL chains, others return None.
    Two collectives: all_gather of record counts, then all_gather of the padded record blocks."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return {c: list(v) for c, v in packets_by_chain.items()}        # single rank: nothing to exchange
    rec = pack_packets(packets_by_chain)
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = torch.device(device) if device is not None else torch.device("cpu")
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([len(rec)], dtype=torch.int64, device=dev))
    most = max(int(c.item()) for c in counts)
    if most == 0:
        return {} if rank == 0 else None
    block = np.zeros(most, dtype=RECORD)
    block[:len(rec)] = rec
    mine = torch.from_numpy(block.view(np.uint8).reshape(-1).copy()).to(dev)
    blocks = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(blocks, mine)
    if rank != 0:
        return None
    parts = [b.cpu().numpy().view(RECORD)[:int(c.item())] for b, c in zip(blocks, counts)]
    return unpack_packets(np.concatenate(parts), chain_names)


_SIDE_STREAM = {}
_GATHER_CAP = {}          # (world, nchains) -> bytes per rank of the exchange buffer; grows when a recording needs more


def gather_rows(rows_by_chain, nchains, names, device=None):
    """The exchange step in table form: every rank's pm_packet rows in, PacketTable (all chains, config order) on rank 0, None
    elsewhere.  = table_from_exchange(exchange_rows(...)); the two halves exist separately so that a pipelined caller can keep the
    collective in one ordered thread and do the indexing / de-dup of rank 0 elsewhere."""
    return table_from_exchange(exchange_rows(rows_by_chain, nchains, device), names)


def exchange_rows(rows_by_chain, nchains, device=None):
    """The collective half.  ONE all_gather per recording in steady state: each rank contributes a fixed-capacity byte block
        int64[2 + nchains]  = payload bytes, rows, rows of each global chain     (always fits)
        payload             = its rows in wire form (40-byte header + len payload bytes each, pm_packets_pack)
    The capacity is agreed without talking: every rank derives it from the headers of the previous exchange, which all ranks
    saw.  If some rank's payload does not fit, every rank sees that in the gathered headers and the exchange is repeated once
    with the capacity they all compute from them.  Ranks other than 0 copy only the headers back from the device.
    -> an opaque value for table_from_exchange (None on ranks other than 0)."""
    import ctypes
    import os
    import torch
    import torch.distributed as dist
    from ._native import check, lib
    from .packet_meta import PacketTable, _stamp
    # PYMODEM_AMD_FORCE_GATHER=1 runs the collective even with one rank (rehearses the RCCL path on a one-GPU box)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not os.environ.get("PYMODEM_AMD_FORCE_GATHER")):
        return ("local", rows_by_chain)
    import time
    trace = os.environ.get("PYMODEM_AMD_GATHER_TRACE")
    tt = [time.perf_counter()]
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = torch.device(device) if device is not None else torch.device("cpu")
    head = 8 * (2 + nchains)
    counts = np.zeros(nchains, dtype=np.int64)
    parts = []
    for c in sorted(rows_by_chain):
        r = rows_by_chain[c]
        if len(r):
            _stamp(r, c)
            counts[c] = len(r)
            parts.append(r)
    mine = PacketTable._stack(parts)
    mine = np.ascontiguousarray(mine)
    need = check(lib().pm_packets_pack(mine.ctypes.data_as(ctypes.c_void_p), len(mine), None, 0))
    key = (world, nchains)
    cap = max(_GATHER_CAP.get(key, 1 << 16), 1 << 12)
    side = None
    if dev.type == "cuda":                                   # copies and the collective on a high-priority stream of their own
        side = _SIDE_STREAM.get(dev)
        if side is None:
            side = _SIDE_STREAM[dev] = torch.cuda.Stream(device=dev, priority=int(os.environ.get("PYMODEM_AMD_EXCHANGE_PRIO", "0")))
    ctxmgr = torch.cuda.stream(side) if side is not None else contextlib.nullcontext()
    with ctxmgr:
        while True:
            block = np.zeros(head + cap, dtype=np.uint8)
            hdr = block[:head].view(np.int64)
            hdr[0], hdr[1], hdr[2:] = need, len(mine), counts
            if need <= cap:
                check(lib().pm_packets_pack(mine.ctypes.data_as(ctypes.c_void_p), len(mine), block[head:].ctypes.data_as(ctypes.c_void_p), cap))
            tt.append(time.perf_counter())
            t = torch.from_numpy(block).to(dev)
            blocks = [torch.empty_like(t) for _ in range(world)]
            tt.append(time.perf_counter())
            dist.all_gather(blocks, t)
            tt.append(time.perf_counter())
            heads = torch.stack([b[:head] for b in blocks]).cpu().numpy()       # every rank needs the headers (capacity agreement)
            hdrs = heads.copy().view(np.int64).reshape(world, 2 + nchains)
            most = int(hdrs[:, 0].max())
            _GATHER_CAP[key] = max(1 << 16, (most + most // 4 + 4095) // 4096 * 4096)       # same value on every rank
            if most <= cap:
                break
            cap = _GATHER_CAP[key]                                               # someone did not fit: once more, with room
    if rank != 0:
        return None
    used = int(hdrs[:, 0].max())
    with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
        got = torch.stack([b[head:head + used] for b in blocks]).cpu().numpy()  # rank 0 only: the payloads, trimmed to the longest
    tt.append(time.perf_counter())
    if trace:
        import sys
        print("[exchange] pack %.2f  h2d %.2f  all_gather %.2f  d2h %.2f ms (%d B/rank)" % (
            *[(b - a) * 1e3 for a, b in zip(tt[:-1], tt[1:])][-4:], head + cap), file=sys.stderr)
    streams = [got[r, :int(hdrs[r, 0])] for r in range(world)]
    return ("streams", streams, hdrs[:, 2:].sum(axis=0).tolist())


class Exchanger:
    """exchange_rows for a stream of recordings, behind: `step(rows)` first collects the gathered blocks of the exchange enqueued
    `depth` (2) collectives before -- steps ago, so the copy back does not wait for the collective to find room on a busy GPU -- then, once
    `batch` recordings have come in, enqueues ONE all_gather for them, and returns a Future of what exchange_rows would have
    returned.  `flush()` resolves what is left.  Call both from ONE thread, in the same order on every rank: every collective,
    including the repeat after a capacity miss (decided from headers all ranks see), is issued there.  `batch` (default 1,
    PYMODEM_AMD_EXCHANGE_BATCH): several recordings per collective.  Measured with a forced one-rank exchange in the pipelined
    executor: the ordered stage costs ~1 ms per RECORDING whether they go one, two or four to a collective (0.9 / 2.5 / 3.6 ms per
    call) -- it is the host's share of a 16-core quota that the packing, indexing and de-dup of the exchange load further, not the
    number of calls -- so the default stays at the lowest latency."""

    def __init__(self, nchains, device=None, batch=None):
        import os
        self.nchains, self.device = nchains, device
        self.batch = max(1, int(batch or os.environ.get("PYMODEM_AMD_EXCHANGE_BATCH", 1)))
        # how many enqueued collectives may be outstanding before the oldest is collected: with one, step() k waited for the
        # collective of step k - 1 -- a tiny kernel that has to find room on a GPU kept full by 14 000-workgroup FIR launches -- for
        # 2 ms per call (Event.synchronize in an all-thread profile); with two it has had a whole further step to finish
        self.depth = max(1, int(os.environ.get("PYMODEM_AMD_EXCHANGE_DEPTH", 2)))
        self._pending = []                     # [(futures, state of an enqueued collective)], oldest first
        self._waiting = []                     # (future, rows) not yet enqueued

    def prepare(self, rows_by_chain):
        """What step() takes: the rows packed for the wire when an exchange will really happen (any thread), the rows themselves
        otherwise."""
        import os
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not os.environ.get("PYMODEM_AMD_FORCE_GATHER")):
            return rows_by_chain
        return pack_rows(rows_by_chain, self.nchains, pinned=self.device is not None and str(self.device).startswith("cuda"))

    def step(self, rows_by_chain):
        import os
        from concurrent.futures import Future
        import torch.distributed as dist
        fut = Future()
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not os.environ.get("PYMODEM_AMD_FORCE_GATHER")):
            fut.set_result(("local", rows_by_chain.rows if isinstance(rows_by_chain, PackedRows) else rows_by_chain))
            return fut
        self._waiting.append((fut, rows_by_chain))
        if len(self._waiting) >= self.batch:
            self._issue()
        return fut

    def _issue(self):
        waiting, self._waiting = self._waiting, []
        try:
            while len(self._pending) >= self.depth:
                self._collect()
            self._pending.append(([f for f, _ in waiting], _exchange_issue([r for _, r in waiting], self.nchains, self.device)))
        except BaseException as e:
            for f, _ in waiting:
                if not f.done():
                    f.set_exception(e)

    def flush(self):
        if self._waiting:
            self._issue()
        while self._pending:
            self._collect()

    def _collect(self):
        """The oldest outstanding collective's results to their futures (every rank collects in the same order: a repeat after a
        capacity miss, decided from headers all ranks see, is a collective of its own)."""
        if not self._pending:
            return
        futs, state = self._pending.pop(0)
        try:
            for f, res in zip(futs, _exchange_collect(state)):
                f.set_result(res)
        except BaseException as e:
            for f in futs:
                if not f.done():
                    f.set_exception(e)


class PackedRows:
    """A rank's rows of one recording in wire form (pack_rows)."""
    __slots__ = ("rows", "counts", "nrows", "payload", "block", "block_cap")


def pack_rows(rows_by_chain, nchains, pinned=False):
    """{global chain index: pm_packet rows} -> PackedRows: the wire form (40-byte header + len payload bytes per row,
    pm_packets_pack) ready for Exchanger.step.  Any thread may do this; only step() has to keep the ranks' order."""
    import ctypes
    from ._native import check, lib
    from .packet_meta import PacketTable, _stamp
    p = PackedRows()
    p.rows, p.counts = rows_by_chain, np.zeros(nchains, dtype=np.int64)
    parts = []
    for c in sorted(rows_by_chain):
        r = rows_by_chain[c]
        if len(r):
            _stamp(r, c)
            p.counts[c] = len(r)
            parts.append(r)
    mine = np.ascontiguousarray(PacketTable._stack(parts))
    p.nrows = len(mine)
    # straight into the fixed-capacity wire block (header | payload) if the capacity the ranks agreed on holds it: the ordered
    # thread then has nothing to copy
    head = 8 * (2 + nchains)
    cap = _wire_cap(nchains)
    p.block = p.block_cap = None
    rows_p = mine.ctypes.data_as(ctypes.c_void_p)
    if cap is not None:
        # one call: it returns the size it needs and has written the rows if that is within the capacity
        from .device import _host_block
        block = _pinned_get(head + cap) if (pinned or _FORCE_PINNED_POOL) else _host_block(head + cap)[:head + cap]
        need = check(lib().pm_packets_pack(rows_p, len(mine), block[head:].ctypes.data_as(ctypes.c_void_p), cap)) if len(mine) else 0
        if need <= cap:
            p.block, p.block_cap = block, cap
            hdr = block[:head].view(np.int64)
            hdr[0], hdr[1], hdr[2:] = need, p.nrows, p.counts
            p.payload = block[head:head + need]
            return p
        _pinned_put(block)                                   # did not fit: nothing refers to it
    need = check(lib().pm_packets_pack(rows_p, len(mine), None, 0)) if len(mine) else 0
    p.payload = np.empty(need, dtype=np.uint8)
    if need:
        check(lib().pm_packets_pack(rows_p, len(mine), p.payload.ctypes.data_as(ctypes.c_void_p), need))
    return p


def _wire_cap(nchains):
    """The exchange capacity per rank currently agreed on (None before the process group exists)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    return max(_GATHER_CAP.get((dist.get_world_size(), nchains), 1 << 16), 1 << 12)


def prewarm_wire_blocks(nchains, count=24, batch=1):
    """Make sure `count` page-locked wire blocks of the capacity currently agreed on are in the pool (and `count // 4` of a batch's
    size): a run that packs more recordings ahead of the ordered thread than any run before it would otherwise register new blocks
    with the runtime in mid-stream -- about a millisecond each, seen as 20 ms outliers on 20-step runs.  Call it after a warm-up."""
    cap = _wire_cap(nchains)
    if cap is None:
        return
    head = 8 * (2 + nchains)
    for size, n in ((head + cap, count), (batch * (head + cap), count // 4 if batch > 1 else 0)):
        got = [_pinned_get(size) for _ in range(n)]
        for blk in got:
            _pinned_put(blk)


def _exchange_issue(packed_list, nchains, device, cap=None):
    """Enqueue ONE all_gather carrying this rank's fixed-capacity blocks (header + packed rows) of len(packed_list) recordings;
    nothing waits for the GPU."""
    import os
    import torch
    import torch.distributed as dist
    if not isinstance(packed_list, (list, tuple)):
        packed_list = [packed_list]
    packed_list = [p if isinstance(p, PackedRows) else pack_rows(p, nchains) for p in packed_list]
    k_rec = len(packed_list)
    world = dist.get_world_size()
    dev = torch.device(device) if device is not None else torch.device("cpu")
    head = 8 * (2 + nchains)
    key = (world, nchains)
    if cap is None:
        cap = max(_GATHER_CAP.get(key, 1 << 16), 1 << 12)
    side = None
    if dev.type == "cuda":
        # A stream of the exchange's own, at NORMAL priority (PYMODEM_AMD_EXCHANGE_PRIO=-1: highest).  Round 2 ran it at the highest so
        # that the tiny collective got onto a full GPU sooner; with the executor's slicer streams at that priority too the exchange's
        # first call of every step -- whatever it was: torch's copy_, a stream query, a plain kernel launch -- blocked the ordered thread
        # for 1.7-1.9 ms (tools/exchange_probe.py: 1.69 ms against 0.02 at normal priority; the interpreter lock was not it: giving it
        # away and taking it back took 5 us there): high-priority streams share a hardware queue.
        side = _SIDE_STREAM.get(dev)
        if side is None:
            side = _SIDE_STREAM[dev] = torch.cuda.Stream(device=dev, priority=int(os.environ.get("PYMODEM_AMD_EXCHANGE_PRIO", "0")))
    from .device import _host_block
    one = head + cap
    # Page-locked blocks and device buffers belong to THIS exchange until _exchange_collect has seen its headers (round 2 handed them
    # out of rings by counter: a block packed for a recording that was still waiting for the ordered thread came round again after
    # 8 or 24 allocations and was overwritten with another recording's rows).  `release` = what goes back to the pools then.
    release = []
    if k_rec == 1 and getattr(packed_list[0], "block", None) is not None and packed_list[0].block_cap == cap:
        block = packed_list[0].block                      # built by pack_rows on a host-stage thread (released with the PackedRows)
    else:
        # recycled host memory (page-locked when it goes to a GPU); bytes past a payload are never read by anyone
        if dev.type == "cuda" or _FORCE_PINNED_POOL:
            block = _pinned_get(k_rec * one)
            release.append(block)
        else:
            block = _host_block(k_rec * one)[:k_rec * one]
        for k, packed in enumerate(packed_list):
            if getattr(packed, "block", None) is not None and packed.block_cap == cap:
                block[k * one:(k + 1) * one] = packed.block
                continue
            need = len(packed.payload)
            hdr = block[k * one:k * one + head].view(np.int64)
            hdr[0], hdr[1], hdr[2:] = need, packed.nrows, packed.counts
            if need <= cap:
                block[k * one + head:k * one + head + need] = packed.payload
    heads_host = done = None
    with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
        # nothing here waits for the GPU, and as few calls as possible -- each one that drops the interpreter lock waits for it
        # again behind a dozen threads (measured in the pipelined executor: 0.4 ms for the copy up, 0.6 for the collective call,
        # 0.3 for the copy of the headers, per call whatever its size; hence several recordings per call, Exchanger.batch).  The
        # block goes up asynchronously (it stays alive in the state), ONE output tensor takes every rank's blocks, and the headers
        # come back into page-locked memory behind the collective: the ordered thread only looks at an event later.
        if dev.type == "cuda":
            # device buffers from a ring of sixteen per shape, never from the allocator: a tensor the collective has used on its own
            # stream goes back to the caching allocator only behind an event, and when the cache runs dry in the middle of the
            # pipeline the hipMalloc behind it waits for the device.  Sixteen = the executor's depth: a slot comes round again
            # only after its recording has long left the post stage (which reads the payloads out of `out` on rank 0).
            rk = (dev, world, k_rec, one)
            _tt = _trace_t()
            pair = _dev_get(rk)
            t, out = pair.t, pair.out
            _tt("dev_get")
            t.copy_(torch.from_numpy(block), non_blocking=True)
            _tt("copy_up")
        else:
            _tt = _trace_t()
            t = torch.from_numpy(block)
            out = None
            pair = None
        if hasattr(dist, "all_gather_into_tensor") and dev.type == "cuda":
            dist.all_gather_into_tensor(out.view(-1), t)
            _tt("all_gather")
        else:
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            out = torch.stack(parts)
        if side is not None:
            ring = _HEADS_RING.setdefault((dev, world, head, k_rec), [])
            if len(ring) < 4:
                ring.append((torch.empty((world, k_rec, head), dtype=torch.uint8, pin_memory=True), torch.cuda.Event()))
                heads_host, done = ring[-1]
            else:
                heads_host, done = ring[_HEADS_SEQ[0] % 4]
            _HEADS_SEQ[0] += 1
            heads_host.copy_(out.view(world, k_rec, one)[:, :, :head], non_blocking=True)
            _tt("heads_down")
            done.record(side)
            _tt("event")
    return {"rows": packed_list, "nchains": nchains, "device": device, "cap": cap, "head": head, "out": out, "side": side, "key": key,
            "keep": block, "heads_host": heads_host, "done": done, "release": release, "pair": pair}


_ISSUE_TRACE = {}


def _trace_t():
    """PYMODEM_AMD_GATHER_TRACE=2: wall time of each call inside _exchange_issue, summed per name (printed by whoever asks)."""
    import os
    import time
    if os.environ.get("PYMODEM_AMD_GATHER_TRACE") != "2":
        return lambda name: None
    last = [time.perf_counter()]

    def mark(name):
        now = time.perf_counter()
        e = _ISSUE_TRACE.setdefault(name, [0, 0.0])
        e[0] += 1
        e[1] += now - last[0]
        last[0] = now
    return mark


_FORCE_PINNED_POOL = False          # tests: use the page-locked pool's bookkeeping without a GPU (plain memory)
_PIN_FREE, _PIN_OWNED, _PIN_LOCK = {}, {}, __import__("threading").Lock()


def _pinned_get(nbytes):
    """A page-locked host block of nbytes that belongs to the caller until _pinned_put.  A copy to the device out of pageable memory is
    staged by the runtime in pieces, each of which the calling thread waits for on a busy GPU; out of page-locked memory it is one
    asynchronous transfer.  The blocks are ordinary (cached) memory registered with the runtime once (pm_host_pin), not memory the
    runtime allocated page-locked: the CPU writes and reads the latter at ~1 GB/s here (2.5 ms to put four half-megabyte blocks
    together), and the packing and the headers are CPU work.  Blocks live for the process's life and go round through a free list per
    size: in steady state nothing is pinned or unpinned (that costs milliseconds each)."""
    import ctypes
    from ._native import lib
    from .device import Context
    with _PIN_LOCK:
        free = _PIN_FREE.setdefault(nbytes, [])
        if free:
            return free.pop()
    blk = np.zeros(nbytes, dtype=np.uint8)
    if not _FORCE_PINNED_POOL:
        try:
            lib().pm_host_pin(Context.default().handle, blk.ctypes.data_as(ctypes.c_void_p), blk.nbytes)    # for the process's life
        except Exception:                                 # noqa: BLE001  (no GPU: plain memory will do)
            pass
    with _PIN_LOCK:
        _PIN_OWNED[id(blk)] = blk
    return blk


def _pinned_put(blk):
    """Hand a block back (no-op for memory that is not the pool's)."""
    if blk is None:
        return
    with _PIN_LOCK:
        if _PIN_OWNED.get(id(blk)) is blk:
            free = _PIN_FREE.setdefault(blk.nbytes, [])
            if not any(b is blk for b in free):
                free.append(blk)


class _DevPair:
    """The device side of one exchange: the rank's block and the gathered blocks of all ranks.  Out of a free list per shape, never
    from the allocator in steady state (a tensor the collective has used on its own stream goes back to the caching allocator only
    behind an event, and when the cache runs dry in the middle of the pipeline the hipMalloc behind it waits for the device).  It
    belongs to its exchange until every reader is done: the ordered thread (headers) and, on rank 0, the post stage of each of its
    recordings (payloads) -- `users` counts them down."""
    __slots__ = ("key", "t", "out", "users", "lock")

    def release(self):
        with self.lock:
            self.users -= 1
            last = self.users == 0
        if last:
            with _PIN_LOCK:
                _DEV_FREE.setdefault(self.key, []).append(self)


_DEV_FREE = {}


def _dev_get(rk):
    import threading
    import torch
    with _PIN_LOCK:
        free = _DEV_FREE.setdefault(rk, [])
        pair = free.pop() if free else None
    if pair is None:
        dev, world, k_rec, one = rk
        pair = _DevPair()
        pair.key, pair.lock = rk, threading.Lock()
        pair.t = torch.empty(k_rec * one, dtype=torch.uint8, device=dev)
        pair.out = torch.empty((world, k_rec * one), dtype=torch.uint8, device=dev)
    pair.users = 1                                        # the exchange itself; _exchange_collect adds rank 0's readers
    return pair


_HEADS_RING, _HEADS_SEQ = {}, [0]       # four page-locked header blocks + events per (device, world, batch): one is in use for a step or two


def _exchange_collect(state):
    """Headers (every rank) and payloads (rank 0) of an enqueued exchange, one result per recording in it; a recording whose payload
    did not fit the agreed capacity on some rank is exchanged again, on its own and synchronously."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    head, cap, out, side, nchains = state["head"], state["cap"], state["out"], state["side"], state["nchains"]
    k_rec, one = len(state["rows"]), state["head"] + state["cap"]
    if state.get("done") is not None:
        if not state["done"].query():                     # enqueued a step or more ago: normally long finished
            state["done"].synchronize()
        heads = state["heads_host"].numpy().copy()
    else:
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            heads = out.view(world, k_rec, one)[:, :, :head].cpu().numpy().copy()
    hdrs_all = heads.view(np.int64).reshape(world, k_rec, 2 + nchains)
    most = int(hdrs_all[:, :, 0].max())
    _GATHER_CAP[state["key"]] = max(1 << 16, (most + most // 4 + 4095) // 4096 * 4096)           # same value on every rank
    pair = state.get("pair")
    results = []
    for k in range(k_rec):
        hdrs = np.ascontiguousarray(hdrs_all[:, k, :])
        if int(hdrs[:, 0].max()) > cap:                   # every rank sees this and repeats, in the same order
            sub = _exchange_issue([state["rows"][k]], nchains, state["device"], cap=_GATHER_CAP[state["key"]])
            results.append(_exchange_collect(sub)[0])
            continue
        if rank != 0:
            results.append(None)
            continue
        blocks = [out[r, k * one:(k + 1) * one] for r in range(world)]
        if pair is not None:
            with pair.lock:
                pair.users += 1                           # released by table_from_exchange once the payloads are on the host
        results.append(("blocks", blocks, hdrs, head, side, pair))      # rank 0: the payloads are copied back by table_from_exchange (any thread)
    # The headers have been seen, so the copy up and the collective are complete (the event behind them has fired), and the repeats
    # above -- the last readers of a PackedRows' payload -- are through: the host blocks go back, and the exchange's own hold on the
    # device buffers ends (rank 0's post stages still hold theirs).
    for blk in state.get("release", ()):
        _pinned_put(blk)
    for packed in state["rows"]:
        blk = getattr(packed, "block", None)
        if blk is not None:
            packed.block = packed.block_cap = packed.payload = None
            _pinned_put(blk)
    if pair is not None:
        pair.release()
    return results


def _streams_from_blocks(blocks, hdrs, head, side, pair=None):
    import torch
    world = len(blocks)
    used = int(hdrs[:, 0].max())
    try:
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            got = torch.stack([b[head:head + used] for b in blocks]).cpu().numpy()
    finally:
        if pair is not None:
            pair.release()                                # the payloads are on the host: this reader is done with the device buffers
    return [got[r, :int(hdrs[r, 0])] for r in range(world)], hdrs[:, 2:].sum(axis=0).tolist()


def table_from_exchange(x, names):
    """The local half on rank 0: index the gathered wire streams into a PacketTable of record heads (payloads stay where they are)."""
    from .packet_meta import PacketTable, _stamp
    if x is None:
        return None
    if x[0] == "local":
        return PacketTable(x[1], names)
    if x[0] == "blocks":
        streams, counts = _streams_from_blocks(*x[1:])
        return PacketTable.from_streams(streams, counts, names)
    return PacketTable.from_streams(x[1], x[2], names)


def correlate(packets_by_chain, nchains, address_distance):
    """De-dup on rank 0: chains fed to Correlate in CONFIG order whatever rank produced them."""
    arr = PacketMetaArray()
    for c in range(nchains):
        arr.add(packets_by_chain.get(c, []))
    arr.CalcCRCs()
    arr.Correlate(address_distance=address_distance)
    return arr
