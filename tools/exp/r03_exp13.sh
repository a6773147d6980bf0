#!/bin/bash
O=gpurun_out/r03e13; mkdir -p $O
SH_BENCH_TEST_CAPTURE_FAIL=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_capfail.json 2>$O/bench_capfail.err
echo "rc=$?"; tail -c 800 $O/bench_capfail.err; cut -c1-900 $O/bench_capfail.json
SH_BENCH_TEST_CAPTURE_FAIL=1 SH_BENCH_FORCE_REDUCER=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_capfail_red.json 2>$O/bench_capfail_red.err
echo "rc=$?"; tail -c 800 $O/bench_capfail_red.err; cut -c1-900 $O/bench_capfail_red.json
SH_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_gloo2.json 2>$O/bench_gloo2.err
echo "rc=$?"; tail -c 500 $O/bench_gloo2.err; cut -c1-900 $O/bench_gloo2.json
SH_BENCH_FORCE_REDUCER=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_red.json 2>$O/bench_red.err
echo "rc=$?"; cut -c1-900 $O/bench_red.json
