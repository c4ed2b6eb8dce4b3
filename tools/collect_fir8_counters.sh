#!/bin/bash
# Run on the GPU box: counter passes over tools/fir8_probe.py (fir8_kernel alone at the engine's shapes), each in its own run, counters only.
cd $GRAFT_REPO_ROOT
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/fir8_counters
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PROBE_NO_EXACT=1 PROBE_REPS=2
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_I8" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pass$i -- python3 $ROOT/tools/fir8_probe.py > $OUT/pass$i.json 2> $OUT/pass$i.err || echo "pass $i ($SET) failed"
done
cd $ROOT
python3 - <<'PY'
import collections, csv, glob, json
out = {}
for f in sorted(glob.glob("gpurun_out/fir8_counters/pass*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "fir8_kernel" not in name:
            continue
        key = ("J4 " if "<4>" in name else "J1 ") + r["Counter_Name"]
        agg[key][0] += 1
        agg[key][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        out[k] = {"launches": n, "avg_per_launch": v / n}
json.dump(out, open("gpurun_out/fir8_counters/summary.json", "w"), indent=1)
for k in sorted(out):
    print(k, round(out[k]["avg_per_launch"]))
PY
