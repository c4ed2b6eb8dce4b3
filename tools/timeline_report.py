#!/usr/bin/env python3
"""Per-recording stage timeline of a pipelined bench run, as a table.

    BENCH_TIMELINE=1 [PYMODEM_AMD_PIPE_WATCH=1] python bench.py --gpus 1 --steps 20 --warmup 5 --also 0 --no-cpu-baseline 2> run.err
    python tools/timeline_report.py run.err

One row per recording of the LAST pipelined call in the file (the timed one): when it was submitted, when its demod finished on the
GPU (`dd`, only with PYMODEM_AMD_PIPE_WATCH=1), when a slicer worker started and ended the batch it went into, how long the host
stage waited for the worker's copy, the host stage's duration, and when finish and post were done -- all in ms since the first
submit.  Batches show up as recordings with the same start: three workers that each took every third recording, a batch stuck for
60 ms behind an allocation, and a garbage collection stalling everything at once were all found by reading this table."""
import ast
import sys


def main(path):
    runs, cur = [], []
    for line in open(path, errors="replace"):
        if not line.startswith("[timeline]"):
            continue
        if "end of run_steps" in line:
            runs.append((cur, float(line.split()[-1])))
            cur = []
        elif "{" in line:
            cur.append(ast.literal_eval(line[line.index("{"):]))
    if not runs:
        sys.exit("no [timeline] lines: run bench.py with BENCH_TIMELINE=1 and keep its stderr")
    recs, end = runs[-2] if len(runs) >= 2 and len(runs[-1][0]) <= 1 else runs[-1]
    print(f"{'rec':>4} {'submit':>13} {'dd':>7} {'slice0':>7} {'slice1':>7} {'fetch+':>7} {'host':>6} {'finish':>7} {'post':>7}")
    for i, d in enumerate(recs):
        f = lambda k: f"{d[k]:7.2f}" if d.get(k) is not None else "      -"
        fetch = d.get("fetched", 0) - d.get("slice1", 0) if "fetched" in d else float("nan")
        host = d.get("host1", 0) - d.get("host0", 0) if "host1" in d else float("nan")
        print(f"{i:4d} {d['submit0']:6.2f}-{d['submit1']:6.2f} {f('demod_done')} {f('slice0')} {f('slice1')} {fetch:7.2f} {host:6.2f} {f('finish1')} {f('post1')}")
    print(f"end of the call: {end:.2f} ms")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/dev/stdin")
