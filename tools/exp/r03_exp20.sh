#!/bin/bash
set -u
O=gpurun_out/r03e20; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv_kernels_vs_oracle or launch_table or golden or stack" > $O/tests_parity.txt 2>&1
grep -E "passed|failed" $O/tests_parity.txt | tail -2
for g in 1 0 1 0; do
  SH_GG_IN3=$g timeout 300 python bench.py --steps 200 --warmup 20 --no-secondary --no-roofline 2> $O/bench_in3_$g.err | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in3=$g', j['ms_per_step'])"
done
python tools/layer_report.py > $O/layer_report.txt 2>&1 || true
grep -E "in3|out3" $O/layer_report.txt; tail -1 $O/layer_report.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests_all.txt 2>&1; grep -E "passed|failed" $O/tests_all.txt | tail -2
