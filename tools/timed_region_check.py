#!/usr/bin/env python3
"""Does the kernel trace agree with the bench line's HIP events?  usage: timed_region_check.py <rocprof dir> <bench line json> <out json>
rocprofv3 --kernel-trace of one bench.py run gives every launch of the dominant kernel with its start and end; the run's own line gives
the kernel's average duration over the TIMED steps from HIP events on its stream (roofline.avg_kernel_ms).  The trace's launches in start
order are warm-up, timed steps, then the line's other legs (exchange, upload, the kernel alone): the timed ones are launches
warmup+1 .. warmup+steps.  The --stats average of the whole run mixes all of those legs and is NOT the figure to compare."""
import csv
import glob
import json
import sys

d, line_path, out_path = sys.argv[1:4]
line = json.load(open(line_path))
kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
want = "afsk_fused8_kernel" if line["roofline"]["kernel"] == "fir_f64" else None
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(kt)) if want and want in r["Kernel_Name"]]
rows.sort()
w, k = line["warmup"], line["steps"]
timed = rows[w:w + k]
dur = [(e - s) / 1e6 for s, e in timed]
span = (timed[-1][1] - timed[0][0]) / 1e6
out = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py " + " ".join(sys.argv[4:]),
       "kernel": want, "launches_in_trace": len(rows), "warmup": w, "steps": k,
       "trace_avg_ms_timed_launches": round(sum(dur) / len(dur), 5), "trace_min_ms": round(min(dur), 5), "trace_max_ms": round(max(dur), 5),
       "trace_avg_ms_all_launches": round(sum((e - s) / 1e6 for s, e in rows) / len(rows), 5),
       "hip_events_avg_ms_timed_launches": line["roofline"]["avg_kernel_ms"], "line_ms_per_step": line["ms_per_step"],
       "trace_first_start_to_last_end_ms_per_step": round(span / k, 5),
       "agreement": round(sum(dur) / len(dur) / line["roofline"]["avg_kernel_ms"], 4),
       "note": "same process, same launches: the kernel trace's start/end of the timed launches against the HIP events bench.py records around "
               "them on the demod streams.  The --stats average over all launches of the run also counts the exchange leg, the upload leg and the "
               "kernel-alone leg of the same line"}
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out))
