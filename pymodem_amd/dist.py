"""Multi-GPU layer.  demod_chains are independent (they share only the read-only recording, pymodem.py:144-149),
so they are sharded over ranks with NO data-path collective; the one exchange step is the gather of decoded-packet
records to rank 0 for the cross-chain de-dup (PacketMetaArray.Correlate, packet_meta.py:230-271), which in the
reference is a multiprocessing.Queue (pymodem.py:140,157-163).  One process per GPU, torch.distributed:
backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.  Payload is KBs: latency-bound.
"""
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


def gather_rows(rows_by_chain, nchains, names, device=None):
    """Table form of gather_packets: pm_packet rows in, PacketTable (all chains, config order) on rank 0, None elsewhere.
    The rows travel as they are (header + the longest packet's bytes), no per-packet Python work."""
    import torch
    import torch.distributed as dist
    from ._native import packet_dtype
    from .packet_meta import PacketTable
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return PacketTable(rows_by_chain, names)
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = torch.device(device) if device is not None else torch.device("cpu")
    dt = packet_dtype()
    mine = []
    for c in sorted(rows_by_chain):
        r = rows_by_chain[c].copy()
        r["source_decoder"] = c
        mine.append(r)
    mine = np.concatenate(mine) if mine else np.zeros(0, dtype=dt)
    width = 40 + (int(mine["len"].max()) if len(mine) else 0)               # header + longest payload
    meta = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(meta, torch.tensor([len(mine), width], dtype=torch.int64, device=dev))
    most, wide = max(int(m[0]) for m in meta), max(int(m[1]) for m in meta)
    if most == 0:
        return PacketTable({}, names) if rank == 0 else None
    block = np.zeros((most, wide), dtype=np.uint8)
    if len(mine):
        block[:len(mine)] = mine.view(np.uint8).reshape(len(mine), dt.itemsize)[:, :wide]
    t = torch.from_numpy(block.reshape(-1)).to(dev)
    blocks = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(blocks, t)
    if rank != 0:
        return None
    by_chain = {}
    for b, m in zip(blocks, meta):
        k = int(m[0])
        if not k:
            continue
        full = np.zeros((k, dt.itemsize), dtype=np.uint8)
        full[:, :wide] = b.cpu().numpy().reshape(most, wide)[:k]
        rows = full.reshape(-1).view(dt)
        for c in np.unique(rows["source_decoder"]):
            by_chain[int(c)] = rows[rows["source_decoder"] == c]
    return PacketTable(by_chain, names)


def correlate(packets_by_chain, nchains, address_distance):
    """De-dup on rank 0: chains fed to Correlate in CONFIG order whatever rank produced them."""
    arr = PacketMetaArray()
    for c in range(nchains):
        arr.add(packets_by_chain.get(c, []))
    arr.CalcCRCs()
    arr.Correlate(address_distance=address_distance)
    return arr
