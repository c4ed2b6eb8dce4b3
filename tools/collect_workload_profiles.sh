#!/bin/bash
# Run on the GPU box: gpurun -- 'bash tools/collect_workload_profiles.sh fsk_9600 [bpsk_300 qpsk_2400 ...]'
# For every workload named: rocprofv3 kernel-trace statistics of its bench command, then the two HBM counter passes (FETCH_SIZE,
# WRITE_SIZE: separate runs, counters only), under gpurun_out/wl/<workload>/.  tools/summarize_profiles.py <tag> <workload> turns
# them into profiles/<tag>_<workload>_{kernel_stats.csv,pmc.json}.  BENCH_ARGS_<workload> overrides the bench arguments.
cd $GRAFT_REPO_ROOT
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for WL in "$@"; do
  OUT=$ROOT/gpurun_out/wl/$WL
  rm -rf $OUT && mkdir -p $OUT
  var=BENCH_ARGS_$WL
  ARGS=${!var:-"--workload $WL --no-cpu-baseline --also 0"}
  PMC_ARGS=${PMC_ARGS:-"--steps 3 --warmup 1"}
  echo "[$WL] kernel trace: $ARGS"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py $ARGS > $OUT/stats.json 2> $OUT/stats.err || { tail -5 $OUT/stats.err; exit 1; }
  echo "[$WL] FETCH_SIZE"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $ARGS $PMC_ARGS > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || { tail -5 $OUT/pmc_fetch.err; exit 1; }
  echo "[$WL] WRITE_SIZE"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $ARGS $PMC_ARGS > $OUT/pmc_write.json 2> $OUT/pmc_write.err || { tail -5 $OUT/pmc_write.err; exit 1; }
  # keep what is merged back small: the per-dispatch traces are tens of megabytes
  find $OUT -name '*_kernel_trace.csv' -delete
  find $OUT -name '*_agent_info.csv' -delete
done
