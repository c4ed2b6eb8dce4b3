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
            d = bytes(bytearray(int(b) & 0xFF for b in p.data[:PKT_BYTES]))
            rec[k]["streamaddress"], rec[k]["chain"], rec[k]["len"] = p.streamaddress, chain, len(d)
            rec[k]["bytes_corrected"] = p.BytesCorrected
            rec[k]["data"][:len(d)] = np.frombuffer(d, dtype=np.uint8)
            k += 1
    return rec


def unpack_packets(rec, chain_names):
    """records -> {chain index: list[PacketMeta]} (decode order within a chain is preserved)."""
    out = {}
    for r in rec:
        p = PacketMeta()
        p.data = r["data"][:int(r["len"])].tolist()
        p.streamaddress = int(r["streamaddress"])
        p.BytesCorrected = int(r["bytes_corrected"])
        p.SourceDecoder = chain_names[int(r["chain"])]
        out.setdefault(int(r["chain"]), []).append(p)
    return out


def gather_packets(packets_by_chain, chain_names, device=None):
    """All ranks call this once per recording.  Rank 0 returns {chain: packets} for ALL chains, others return None.
    Two collectives: all_gather of record counts, then all_gather of the padded record blocks."""
    import torch
    import torch.distributed as dist
    rec = pack_packets(packets_by_chain)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return unpack_packets(rec, chain_names)
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


def correlate(packets_by_chain, nchains, address_distance):
    """De-dup on rank 0: chains fed to Correlate in CONFIG order whatever rank produced them."""
    arr = PacketMetaArray()
    for c in range(nchains):
        arr.add(packets_by_chain.get(c, []))
    arr.CalcCRCs()
    arr.Correlate(address_distance=address_distance)
    return arr
