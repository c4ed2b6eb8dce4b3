#!/usr/bin/env python3
"""How the carrier-loop batch engine scales with recordings in flight (pm_lbatch): engine alone (audio -> sign bitmaps) and the
whole path (engine -> slicers -> LFSR + codec) for bpsk_300 (1 chain) and qpsk_2400 (8 chains) at BASELINE size.  Run on the GPU
box; prints one JSON line per measurement.

    python tools/loopbatch_bench.py [workload ...]        LB_R="8,32" LB_N=28800000 LB_CHUNK=0 LB_FULL=1
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pymodem_amd  # noqa: E402
from pymodem_amd import chain_builder as cb  # noqa: E402
from pymodem_amd.loop_batch import LoopBatch, process_recordings_device  # noqa: E402

N = int(os.environ.get("LB_N", 28_800_000))
RS = [int(v) for v in os.environ.get("LB_R", "8,32,64,128").split(",")]
CHUNK = int(os.environ.get("LB_CHUNK", 0))
FULL = os.environ.get("LB_FULL", "1") == "1"
ctx = pymodem_amd.Context.default(0)


class A:
    pass


for wl in (sys.argv[1:] or ["bpsk_300", "qpsk_2400"]):
    factory, cpg, _ = bench.WORKLOADS[wl]
    args = A()
    args.samples, args.buffer, args.rate, args.workload = N, "signal", 48000, wl
    audio = bench.make_buffer(args)
    d_audio = ctx.upload(audio)
    lines = [factory(c) for c in range(cpg)]
    modems = [cb.ModemConfigurator(48000, ln["modem"]) for ln in lines]
    for R in RS:
        eng = LoopBatch(modems, R, ctx, CHUNK)
        nout, lc, chunks = eng.geometry(N)
        eng.run([d_audio] * min(R, 2))                       # first launches, buffers
        ctx.sync()
        t0 = time.perf_counter()
        eng.run([d_audio] * R)
        t_enq = time.perf_counter() - t0
        ctx.sync()
        dt = time.perf_counter() - t0
        row = {"workload": wl, "recordings": R, "chains": cpg, "loops_in_flight": R * cpg, "chunk": lc, "chunks": chunks,
               "engine_s": round(dt, 3), "enqueue_s": round(t_enq, 3), "engine_Msamples_per_s": round(N * R * cpg / dt / 1e6, 1)}
        print(json.dumps(row), flush=True)
        eng.close()
        if FULL:
            def sets():
                out = []
                for _ in range(R):
                    cs = []
                    for ln, m in zip(lines, modems):
                        m.reset()
                        cs.append([ln["object_name"], m, cb.SlicerConfigurator(48000, ln["slicer"]), cb.StreamConfigurator(ln["stream"]),
                                   cb.CodecConfigurator(ln["codec"], ln["object_name"])])
                    out.append(cs)
                return out
            t0 = time.perf_counter()
            rows = process_recordings_device(sets(), [d_audio] * R, ctx, CHUNK, rows=True)
            dt = time.perf_counter() - t0
            pk = sum(len(v) for v in rows[0])
            print(json.dumps({"workload": wl, "recordings": R, "whole_path_s": round(dt, 3), "Msamples_per_s": round(N * R * cpg / dt / 1e6, 1),
                              "packet_rows_recording0": pk}), flush=True)
            from pymodem_amd import loop_batch
            loop_batch.close_engines()
