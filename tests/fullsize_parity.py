#!/usr/bin/env python3
"""One-off end-to-end check at BASELINE size: every chain of a bench workload over the full 28.8 M-sample bench buffer, GPU path
(group executor) against the oracle (canonical FIR order): slicer bytes, addresses and packets must be identical.
    python tests/fullsize_parity.py [workload] [samples]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "afsk_1200_super_opt"
    samples = int(sys.argv[2]) if len(sys.argv) > 2 else 28_800_000

    class A:
        pass
    args = A()
    args.samples, args.rate, args.workload, args.buffer = samples, 48000, workload, "signal"
    audio = bench.make_buffer(args)
    factory, cpg, _ = bench.WORKLOADS[workload]
    lines = [factory(c) for c in range(cpg)]
    from pymodem_amd import chain_builder as cb, chain_execute as ce
    chains = [cb.build_chain(48000, l) for l in lines]
    stages = {}
    t0 = time.perf_counter()
    pk = ce.process_chains_device(chains, audio, stages=stages)
    t_gpu = time.perf_counter() - t0
    report = {"workload": workload, "samples": samples, "chains": [], "gpu_seconds": round(t_gpu, 3)}
    ok = True
    for c, line in enumerate(lines):
        t0 = time.perf_counter()
        r = O.run_chain(O.build_chain(48000, line), audio, canon=True)
        sl = stages["sliced"][c]
        same_bytes = bool(np.array_equal(sl.data, r["slice_data"]) and np.array_equal(sl.address, r["slice_addr"]))
        got = [(p.streamaddress, bytes(bytearray(p.data)), p.BytesCorrected) for p in pk[c]]
        want = [(p.streamaddress, bytes(bytearray(p.data)), p.BytesCorrected) for p in r["packets"]]
        same_pk = got == want
        ok &= same_bytes and same_pk
        report["chains"].append({"chain": line["object_name"], "slicer_bytes": int(len(sl.data)), "slicer_identical": same_bytes,
                                 "packets": len(want), "packets_identical": same_pk, "oracle_seconds": round(time.perf_counter() - t0, 2)})
    report["all_identical"] = bool(ok)
    print(json.dumps(report))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
