#!/bin/bash
# Run on the GPU box: SQ / GRBM counters per launch of the native executor's kernels (three passes, counters only) over a short run
# of the headline bench -> gpurun_out/pipeline_counters/summary.json (copied to profiles/<round>_pipeline_kernel_counters.json).
cd $GRAFT_REPO_ROOT
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/pipeline_counters
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pass$i -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --also 0 --with-exchange 0 > $OUT/pass$i.json 2> $OUT/pass$i.err || echo "pass $i failed"
done
cd $ROOT
python3 - <<'PY'
import collections, csv, glob, json, re
out = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/pipeline_counters/pass*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel)(<[^>(]*>)?", r["Kernel_Name"])
        if not m:
            continue
        k = m.group(1) + (m.group(2) or "")
        agg[(k, r["Counter_Name"])][0] += 1
        agg[(k, r["Counter_Name"])][1] += float(r["Counter_Value"])
    for (k, c), (n, v) in agg.items():
        out[k][c] = {"launches": n, "avg": v / n}
json.dump(out, open("gpurun_out/pipeline_counters/summary.json", "w"), indent=1)
for k in sorted(out):
    if any(s in k for s in ("lpf8", "bpf8", "fused8", "sweep_exact", "slice_walk", "slice_pack")):
        print(k, {c: round(v["avg"]) for c, v in sorted(out[k].items())})
PY
