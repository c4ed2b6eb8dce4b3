#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/collect_profiles.sh'): the default bench line, rocprofv3 kernel-trace stats of the same
# command, and the two PMC passes (separate runs, counters only), written under gpurun_out/; tools/summarize_profiles.py then
# turns them into profiles/*.
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/prof_stats $OUT/pmc_fetch $OUT/pmc_write
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --also 0 > $OUT/prof_stats.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --also 0 > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --also 0 > $OUT/pmc_write.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
KB_REPS=5 python tools/kernel_bench.py > $OUT/kernel_bench.jsonl 2>/dev/null
python bench.py --workload fsk_9600 --no-cpu-baseline --also 0 > $OUT/bench_fsk_9600.json 2> $OUT/bench_fsk_9600.err
python bench.py --overlap 0 --steps 10 --no-cpu-baseline --also 0 > $OUT/bench_overlap0.json 2> $OUT/bench_overlap0.err
python bench.py --gpus 1 --steps 20 --warmup 5 --also 0 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
cat $OUT/bench_default.json
