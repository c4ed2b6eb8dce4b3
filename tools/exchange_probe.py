#!/usr/bin/env python3
"""Where the ordered thread of the packet exchange spends its time: bench.py with dist._exchange_issue / _exchange_collect wrapped in
timers (wall and thread CPU).  Run on the GPU box with PYMODEM_AMD_FORCE_GATHER=1 and the rendezvous variables set."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymodem_amd import dist as pdist  # noqa: E402

acc = {}


def wrap(name):
    real = getattr(pdist, name)

    def timed(*a, **k):
        t0, c0 = time.perf_counter(), time.thread_time()
        try:
            return real(*a, **k)
        finally:
            e = acc.setdefault(name, [0, 0.0, 0.0])
            e[0] += 1
            e[1] += time.perf_counter() - t0
            e[2] += time.thread_time() - c0
    setattr(pdist, name, timed)


for n in ("_exchange_issue", "_exchange_collect", "pack_rows", "table_from_exchange"):
    wrap(n)
import bench  # noqa: E402
import runpy  # noqa: E402

sys.argv = ["bench.py"] + sys.argv[1:]
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
finally:
    for n, (cnt, wall) in pdist._ISSUE_TRACE.items():
        print(f"[exchange probe] inside _exchange_issue, {n}: {wall / max(cnt, 1) * 1e3:.3f} ms per call", file=sys.stderr)
    for n, (cnt, wall, cpu) in acc.items():
        print(f"[exchange probe] {n}: {cnt} calls, {wall / max(cnt, 1) * 1e3:.3f} ms wall, {cpu / max(cnt, 1) * 1e3:.3f} ms thread CPU per call", file=sys.stderr)
