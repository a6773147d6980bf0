#!/bin/bash
O=gpurun_out/r03e36; mkdir -p $O
timeout 900 python -m pytest tests/test_semantic.py tests/test_parallel_gloo.py -x -q -m gpu > $O/tests_sem.txt 2>&1
grep -E "passed|failed|^E  " $O/tests_sem.txt | head -6
timeout 300 python tools/bench_semantic.py --graph --steps 50 2>&1 | tail -1 | cut -c1-260
