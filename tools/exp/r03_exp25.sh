#!/bin/bash
set -u
O=gpurun_out/r03e25; mkdir -p $O
timeout 900 python -m pytest tests/test_semantic.py -x -q -m gpu > $O/tests_sem.txt 2>&1
grep -E "passed|failed|^E " $O/tests_sem.txt | tail -8
timeout 300 python tools/bench_semantic.py --graph --steps 50 2>&1 | tail -1 | cut -c1-300
timeout 300 python tools/exp/skl_diag.py 2>&1 | tail -12
