#!/usr/bin/env python3
"""Kernel (and copy) timeline of a short native-executor run from rocprofv3 CSVs: per queue, when each class of kernel ran during the
timed steps -- to see what the slicer streams wait for while the demod streams are full.  usage: pipe_timeline.py <dir> [t0_ms t1_ms]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
ev = []
for r in csv.DictReader(open(kt)):
    n = r["Kernel_Name"]
    key = ("bpf8" if "bpf8" in n else "lpf8" if "slide_lpf8" in n else "exact" if "sweep_exact" in n else "walk" if "slice_walk" in n else
           "slice_" + n.split("slice_")[1].split("_kernel")[0] if "slice_" in n else "copyk" if "copyBuffer" in n else "fill" if "fillBuffer" in n else n[:24])
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), key, "q" + r.get("Queue_Id", "?")))
for f in glob.glob(d + "/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy " + r.get("Direction", ""), "dma"))
ev.sort()
t_first = ev[0][0]
lp = [e for e in ev if e[2] == "lpf8"]
# the window: from lpf8 launch number argv[2] (two per recording) to number argv[3]
i0, i1 = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (len(lp) - 40, len(lp) - 1)
t0 = lp[i0][0] - 200000
t_end = lp[min(i1, len(lp) - 1)][1] + 15000000
ev = [e for e in ev if e[0] <= t_end]
print("lpf8 launches:", len(lp))
print("queues:", dict(collections.Counter((e[3], e[2]) for e in ev if e[0] >= t0)))
last = {}
for e in ev:
    if e[0] < t0:
        continue
    print(f"{(e[0] - t0) / 1e6:8.3f} {(e[1] - t0) / 1e6:8.3f} {(e[1] - e[0]) / 1e3:9.1f} us  {e[3]:4s} {e[2]}")
