"""Multi-GPU layer on CPU: world size 2, gloo.  Chains are sharded over ranks, every rank contributes its packets, rank 0
de-dups in config order.  Uses the reference's packets of the bundled recording (afsk_300.json: 49 good / 6 bad)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

WORKER = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from pymodem_amd import dist as pdist
from pymodem_amd.packet_meta import PacketMeta
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend, device = os.environ.get("PM_TEST_BACKEND", "gloo"), os.environ.get("PM_TEST_DEVICE") or None
if backend == "nccl":
    import torch
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
else:
    dist.init_process_group("gloo", rank=rank, world_size=world)
g = np.load(os.path.join(sys.argv[1], "tests", "golden", "wav_chains.npz"))
names = json.load(open(os.path.join(sys.argv[1], "tests", "golden", "wav_chains_summary.json")))["afsk_300"]["chains"]
mine = pdist.shard_chains(len(names), rank, world)
pk = {}
for c in mine:
    lens, data, addr, corr = g[f"afsk_300__c{c}_pkt_len"], g[f"afsk_300__c{c}_pkt_data"], g[f"afsk_300__c{c}_pkt_addr"], g[f"afsk_300__c{c}_pkt_corrected"]
    pos, lst = 0, []
    for k in range(len(lens)):
        p = PacketMeta(); p.data = data[pos:pos + lens[k]].tolist(); pos += int(lens[k])
        p.streamaddress, p.BytesCorrected, p.SourceDecoder = int(addr[k]), int(corr[k]), names[c]
        lst.append(p)
    pk[c] = lst
from pymodem_amd._native import packet_dtype
rows = {}
pk_all = dict(pk)
for c in range(len(names)):              # (every rank can build every chain's rows: what rank 0 must end up with is known everywhere)
    if c in pk_all:
        continue
    lens, data, addr, corr = g[f"afsk_300__c{c}_pkt_len"], g[f"afsk_300__c{c}_pkt_data"], g[f"afsk_300__c{c}_pkt_addr"], g[f"afsk_300__c{c}_pkt_corrected"]
    pos, lst = 0, []
    for k in range(len(lens)):
        p = PacketMeta(); p.data = data[pos:pos + lens[k]].tolist(); pos += int(lens[k])
        p.streamaddress, p.BytesCorrected, p.SourceDecoder = int(addr[k]), int(corr[k]), names[c]
        lst.append(p)
    pk_all[c] = lst
for c, lst in pk_all.items():            # the same packets as pm_packet rows (what the native codecs hand out)
    r = np.zeros(len(lst), dtype=packet_dtype())
    for k, p in enumerate(lst):
        p.CalcCRC(); p.Validate()
        r[k]["streamaddress"], r[k]["len"], r[k]["bytes_corrected"] = p.streamaddress, len(p.data), p.BytesCorrected
        r[k]["calculated_crc"], r[k]["carried_crc"], r[k]["valid_crc"], r[k]["valid_header"] = p.CalculatedCRC, p.CarriedCRC, p.ValidCRC, p.ValidHeader
        r[k]["data"][:len(p.data)] = p.data
    rows[c] = r
rows_all = rows
rows = {c: rows_all[c] for c in mine}
pdist._GATHER_CAP[(world, len(names))] = 4096      # too small on purpose: the first exchange must notice and repeat itself
first = pdist.gather_rows({c: r.copy() for c, r in rows.items()}, len(names), names, device)
assert pdist._GATHER_CAP[(world, len(names))] > 4096
table = pdist.gather_rows(rows, len(names), names, device)  # steady state: one collective
if rank == 0:
    assert np.array_equal(first.rows, table.rows) and first.counts == table.counts
    assert table.counts == [len(g[f"afsk_300__c{c}_pkt_len"]) for c in range(len(names))]
    assert np.all(np.diff(table.rows["source_decoder"]) >= 0)
    table.correlate(8000 / 40)
    print("TABLE " + json.dumps({"good": table.CountGood(), "bad": table.CountBad(), "addr": table.rows["streamaddress"][table.unique_idx].tolist(),
          "dec": table.unique_decoders}))
else:
    assert table is None
# the deferred form: three recordings through an Exchanger, the first against a capacity that is too small; every future must give
# what the synchronous exchange gave
for batch, depth in ((1, 1), (2, 1), (1, 2), (2, 2)):
    pdist._GATHER_CAP[(world, len(names))] = 4096
    ex = pdist.Exchanger(len(names), device, batch=batch)
    ex.depth = depth                                                           # collectives outstanding before the oldest is collected
    futs = [ex.step({c: r.copy() for c, r in rows.items()}) for _ in range(3)]
    if batch == 1 and depth == 1:
        assert futs[0].done() and futs[1].done() and not futs[2].done()        # each step resolves the one before it
    elif batch == 1:
        assert futs[0].done() and not futs[1].done() and not futs[2].done()    # ... the one two before it
    else:
        assert not any(f.done() for f in futs)                                 # two recordings per collective: enqueued at the second step
    ex.flush()
    for f in futs:
        t2 = pdist.table_from_exchange(f.result(), names)
        if rank == 0:
            assert np.array_equal(t2.rows, first.rows) and t2.counts == first.counts        # (`table` has been through correlate() since)
        else:
            assert t2 is None
    assert pdist._GATHER_CAP[(world, len(names))] > 4096
# Many recordings packed for the wire BEFORE the ordered thread gets to any of them, across a capacity growth (the first recording
# does not fit the agreed capacity, the blocks of the others were built for the old one): every recording must come out with its own
# rows.  Round 2 handed the page-locked blocks out of a ring by counter and a waiting recording's block came round again (ADVICE r2).
from pymodem_amd.packet_meta import PacketTable
pdist._FORCE_PINNED_POOL = device is None          # without a GPU: the pool's bookkeeping on plain memory
def rec_rows(k, chains):
    out = {}
    for c in chains:
        r = rows_all[c]
        r = r.copy() if k == 0 else r[:(k + c) % 3 + 1].copy()
        r["streamaddress"] += 100000 * k
        out[c] = r
    return out
for batch in (1, 3):
    pdist._GATHER_CAP[(world, len(names))] = 4096
    ex = pdist.Exchanger(len(names), device, batch=batch)
    NREC = 30
    packed = [pdist.pack_rows(rec_rows(k, mine), len(names), pinned=device is not None) for k in range(NREC)]
    assert any(p.block is not None for p in packed[1:])
    futs = [ex.step(p) for p in packed]
    ex.flush()
    for k, f in enumerate(futs):
        t = pdist.table_from_exchange(f.result(), names)
        if rank == 0:
            want = PacketTable(rec_rows(k, range(len(names))), names)
            assert t.counts == want.counts, (k, t.counts, want.counts)
            for field in ("streamaddress", "len", "calculated_crc", "source_decoder"):
                assert np.array_equal(t.heads[field], want.heads[field]), (batch, k, field)
            assert np.array_equal(t.rows["data"], want.rows["data"]), (batch, k)
    free = sum(len(v) for v in pdist._PIN_FREE.values())
    assert free == len(pdist._PIN_OWNED), (free, len(pdist._PIN_OWNED))      # every block went back to the pool
pdist._FORCE_PINNED_POOL = False
for _ in range(2):                       # twice: the exchange must be repeatable
    got = pdist.gather_packets(pk, names, device)
if rank == 0:
    arr = pdist.correlate(got, len(names), 8000 / 40)
    u = arr.unique_packet_array
    tu = table.unique_packets()                 # compact table: payloads come straight out of the gathered wire streams
    assert [bytes(bytearray(p.data)) for p in u] == [bytes(bytearray(p.data)) for p in tu]
    assert [list(p.CorrelatedDecoders) for p in u] == [list(p.CorrelatedDecoders) for p in tu]
    print("RESULT " + json.dumps({"good": arr.CountGood(), "bad": arr.CountBad(), "addr": [p.streamaddress for p in u],
          "dec": [list(p.CorrelatedDecoders) for p in u], "per_chain": {str(c): len(v) for c, v in got.items()}}))
else:
    assert got is None
dist.barrier()
dist.destroy_process_group()
'''


def test_shard_chains_partitions():
    from pymodem_amd.dist import shard_chains
    for n in [1, 5, 8, 64]:
        for w in [1, 2, 3, 8]:
            parts = [shard_chains(n, r, w) for r in range(w)]
            assert sorted(c for p in parts for c in p) == list(range(n))


def test_record_round_trip(golden):
    from pymodem_amd import dist as pdist
    from pymodem_amd.packet_meta import PacketMeta
    rng = np.random.default_rng(3)
    pk = {}
    for c in [4, 0, 2]:
        lst = []
        for _ in range(int(rng.integers(0, 5))):
            p = PacketMeta()
            p.data = rng.integers(0, 256, int(rng.integers(2, 1100))).tolist()
            p.streamaddress, p.BytesCorrected = int(rng.integers(0, 2 ** 40)), int(rng.integers(0, 9))
            lst.append(p)
        pk[c] = lst
    back = pdist.unpack_packets(pdist.pack_packets(pk), [f"chain{c}" for c in range(5)])
    for c, lst in pk.items():
        assert [(p.streamaddress, p.data, p.BytesCorrected) for p in lst] == \
               [(p.streamaddress, p.data, p.BytesCorrected) for p in back.get(c, [])]
        assert all(p.SourceDecoder == f"chain{c}" for p in back.get(c, []))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_multi_rank_gather_and_dedup(tmp_path, golden, world):
    """world 2: ranks interleave the 5 chains; world 4: one rank owns two chains, one of the others none with packets; world 8 (the
    node the scaling bench runs on): three ranks own no chain at all -- the exchange must cope with empty contributions as well.
    Every world also runs 30 recordings that were packed for the wire before any was exchanged, across a capacity growth."""
    g = golden("wav_chains")
    summ = json.load(open(os.path.join(GOLDEN, "wav_chains_summary.json")))["afsk_300"]
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29500 + os.getpid() % 2000
    procs = []
    port += world
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    line = [l for l in outs[0][0].splitlines() if l.startswith("RESULT ")][0]
    res = json.loads(line[7:])
    assert res["good"] == summ["good"] == 49 and res["bad"] == summ["bad"] == 6
    assert res["addr"] == g["afsk_300__uniq_addr"].tolist()
    assert res["dec"] == summ["uniq_decoders"]           # config order, although ranks 0 and 1 interleave the chains
    assert res["per_chain"] == {"0": 5, "2": 48, "3": 47}
    tab = json.loads([l for l in outs[0][0].splitlines() if l.startswith("TABLE ")][0][6:])
    assert tab["good"] == 49 and tab["bad"] == 6 and tab["addr"] == res["addr"] and tab["dec"] == res["dec"]


@pytest.mark.gpu
def test_forced_one_rank_rccl_exchange_on_the_gpu(tmp_path, golden):
    """The same worker with the collectives on the GPU (backend nccl = RCCL, one rank, PYMODEM_AMD_FORCE_GATHER: the exchange runs
    although nobody else is there): page-locked blocks and device buffers out of their pools, 30 recordings packed before any is
    exchanged, a capacity growth in between."""
    summ = json.load(open(os.path.join(GOLDEN, "wav_chains_summary.json")))["afsk_300"]
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29400 + os.getpid() % 500),
               PM_TEST_BACKEND="nccl", PM_TEST_DEVICE="cuda:0", PYMODEM_AMD_FORCE_GATHER="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    tab = json.loads([l for l in p.stdout.splitlines() if l.startswith("TABLE ")][0][6:])
    assert tab["good"] == summ["good"] == 49 and tab["bad"] == summ["bad"] == 6


def test_bench_launches_its_own_ranks_and_relays_their_exit_code():
    """`python bench.py --gpus 2` with no launcher around it starts torch.distributed.run as a child (never an exec, and before this
    process could have touched a GPU).  Without a GPU the ranks fail loudly -- there is no CPU fallback -- and the parent hands the
    failure on instead of hanging or printing a line."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("the GPU box runs the real thing (test_bench_two_ranks_by_itself)")
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--also", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert "needs a GPU" in p.stderr and '"metric"' not in p.stdout


@pytest.mark.gpu
def test_bench_two_ranks_by_itself():
    """The scaling bench's command with N = 2 and nothing around it: two ranks share the one GPU of the test box
    over gloo, rank 0's ONE line comes back through the parent with n_gpus 2, the CPU baseline the parent measured before any rank
    existed, the strong-scaling figure beside the weak one -- and `also.qpsk_2400_64`, BASELINE configs[4]'s own curve point: the 64
    chains of the qpsk sweep divided over the ranks (32 each here), one engine run per rank, the exchange and rank 0's de-dup over
    all 64 behind it (here at rehearsal size -- 12 recordings of 400 000 samples -- the driver's command runs it in full)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BENCH_ALSO64_RECORDINGS="12", BENCH_ALSO64_SAMPLES="400000")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "4", "--warmup", "1",
                        "--cpu-sample", "480000"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["value"] > 0
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port"
    assert d["config"]["chains_total"] == 16
    assert d["strong"]["chains_total"] == 8 and d["strong"]["value"] > 0
    q = d["also"]["qpsk_2400_64"]
    assert q["chains_total"] == 64 and q["chains_per_gpu"] == 32 and q["n_gpus"] == 2 and q["scaling"] == "strong"
    assert q["steps"] == 12 and q["value"] > 0 and q["rehearsal"]["BENCH_ALSO64_SAMPLES"] == 400000
    assert q["packets"] is not None
