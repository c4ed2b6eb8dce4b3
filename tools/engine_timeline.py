#!/usr/bin/env python3
"""Kernel timeline of a batch-engine run from a rocprofv3 --kernel-trace CSV: which of the engine's kernels (band-pass, AGC rows, carrier
loops, matched filters, exact recomputation, copies) ran when and on which queue, for a window in the middle of the run, and how much
of the loops' time the other streams' kernels overlap.  usage: engine_timeline.py <kernel_trace.csv> [window_ms]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
ev = []
for r in rows:
    n = r["Kernel_Name"]
    key = ("agc" if "agc_rows" in n else "loop" if "loop_direct" in n or "loop_kernel" in n else "fir8" if "fir8_kernel" in n else
           "exact" if "fir8_exact" in n else "fir_rows" if "fir_rows" in n else "copy" if "rows_copy" in n else "max" if "rows_max" in n else
           "slice" if "slice" in n else "other")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), key, r.get("Queue_Id", "?")))
ev.sort()
big = [e for e in ev if e[2] == "loop"]
mid = big[len(big) // 2][0]
print("kernels:", dict(collections.Counter(e[2] for e in ev)))
print("queues:", dict(collections.Counter((e[2], e[3]) for e in ev if e[2] in ("agc", "loop", "fir8", "fir_rows", "copy"))))
for e in ev:
    if mid - 0.1 * win * 1e6 < e[0] < mid + 0.9 * win * 1e6 and e[2] != "slice":
        print(f"{(e[0] - mid) / 1e6:9.2f} {(e[1] - mid) / 1e6:9.2f} {(e[1] - e[0]) / 1e6:8.2f} ms  {e[2]:8s} q{e[3]}")
loops = [e for e in big[len(big) // 4: 3 * len(big) // 4]]
span = loops[-1][1] - loops[0][0]
busy = sum(e[1] - e[0] for e in loops)
print(f"middle half of the run: {len(loops)} loop launches, {busy / span:.3f} of the time inside a loop launch, {span / len(loops) / 1e6:.2f} ms per chunk")
