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
    from .packet_meta import PacketTable
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
            r["source_decoder"] = c
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
            side = _SIDE_STREAM[dev] = torch.cuda.Stream(device=dev, priority=-1)
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


def table_from_exchange(x, names):
    """The local half on rank 0: index the gathered wire streams into a PacketTable of record heads (payloads stay where they are)."""
    from .packet_meta import PacketTable
    if x is None:
        return None
    if x[0] == "local":
        return PacketTable(x[1], names)
    return PacketTable.from_streams(x[1], x[2], names)


def correlate(packets_by_chain, nchains, address_distance):
    """De-dup on rank 0: chains fed to Correlate in CONFIG order whatever rank produced them."""
    arr = PacketMetaArray()
    for c in range(nchains):
        arr.add(packets_by_chain.get(c, []))
    arr.CalcCRCs()
    arr.Correlate(address_distance=address_distance)
    return arr
