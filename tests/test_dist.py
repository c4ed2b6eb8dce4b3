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
for c, lst in pk.items():                # the same packets as pm_packet rows (what the native codecs hand out)
    r = np.zeros(len(lst), dtype=packet_dtype())
    for k, p in enumerate(lst):
        p.CalcCRC(); p.Validate()
        r[k]["streamaddress"], r[k]["len"], r[k]["bytes_corrected"] = p.streamaddress, len(p.data), p.BytesCorrected
        r[k]["calculated_crc"], r[k]["carried_crc"], r[k]["valid_crc"], r[k]["valid_header"] = p.CalculatedCRC, p.CarriedCRC, p.ValidCRC, p.ValidHeader
        r[k]["data"][:len(p.data)] = p.data
    rows[c] = r
pdist._GATHER_CAP[(world, len(names))] = 4096      # too small on purpose: the first exchange must notice and repeat itself
first = pdist.gather_rows({c: r.copy() for c, r in rows.items()}, len(names), names)
assert pdist._GATHER_CAP[(world, len(names))] > 4096
table = pdist.gather_rows(rows, len(names), names)  # steady state: one collective
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
for batch in (1, 2):
    pdist._GATHER_CAP[(world, len(names))] = 4096
    ex = pdist.Exchanger(len(names), batch=batch)
    futs = [ex.step({c: r.copy() for c, r in rows.items()}) for _ in range(3)]
    if batch == 1:
        assert futs[0].done() and futs[1].done() and not futs[2].done()        # each step resolves the one before it
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
for _ in range(2):                       # twice: the exchange must be repeatable
    got = pdist.gather_packets(pk, names)
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


@pytest.mark.parametrize("world", [2, 4])
def test_multi_rank_gather_and_dedup(tmp_path, golden, world):
    """world 2: ranks interleave the 5 chains; world 4: one rank owns two chains, one of the others none with packets -- the
    exchange must cope with empty contributions as well."""
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
