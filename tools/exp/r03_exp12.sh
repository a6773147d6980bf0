#!/bin/bash
# round 3, experiment 12: the multi-rank control flow of bench.py with two ranks sharing the one GPU over gloo (the capture
# cannot hold a host-side collective: exercises the logged fall-back to eager launches on every rank), and the new bf16 tests
O=gpurun_out/r03e12; mkdir -p $O
SH_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_gloo2.json 2>$O/bench_gloo2.err
echo "rc=$?"; tail -c 1500 $O/bench_gloo2.err; cat $O/bench_gloo2.json | cut -c1-600
SH_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline --no-graph > $O/bench_gloo2_eager.json 2>$O/bench_gloo2_eager.err
echo "rc=$?"; cat $O/bench_gloo2_eager.json | cut -c1-300
timeout 600 python -m pytest tests/test_bf16.py tests/test_semantic.py -q -m gpu > $O/tests_bf16_sem.txt 2>&1
grep -n "passed\|failed\|FAILED" $O/tests_bf16_sem.txt | tail -5
