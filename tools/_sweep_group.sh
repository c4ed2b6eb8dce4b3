run() { # name workers env...
 n=$1; W=$2; shift; shift
 env "$@" BENCH_TIMELINE=1 python bench.py --gpus 1 --steps 20 --warmup 5 --also 0 --slice-workers $W --no-cpu-baseline > gpurun_out/i_${n}_20.json 2> gpurun_out/i_${n}_20.err || exit 1
 env "$@" python bench.py --gpus 1 --steps 20 --warmup 5 --also 0 --slice-workers $W --no-cpu-baseline > gpurun_out/i_${n}_20b.json 2> gpurun_out/i_${n}_20b.err || exit 1
 env "$@" python bench.py --gpus 1 --steps 20 --warmup 5 --also 0 --slice-workers $W --no-cpu-baseline > gpurun_out/i_${n}_20c.json 2> gpurun_out/i_${n}_20c.err || exit 1
 env "$@" python bench.py --gpus 1 --steps 400 --warmup 10 --also 0 --slice-workers $W --no-cpu-baseline > gpurun_out/i_${n}_400.json 2> gpurun_out/i_${n}_400.err || exit 1
}
run inl_w2g3 2 PYMODEM_AMD_SLICE_MIN_GROUP=3
run cpy_w2g3 2 PYMODEM_AMD_SLICE_MIN_GROUP=3 PYMODEM_AMD_FETCH=copy
run inl_w2g2 2 PYMODEM_AMD_SLICE_MIN_GROUP=2
run inl_w3g2 3 PYMODEM_AMD_SLICE_MIN_GROUP=2
run inl_w2g4 2 PYMODEM_AMD_SLICE_MIN_GROUP=4
