#!/bin/bash
set -u
O=gpurun_out/r03e21; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv_kernels_vs_oracle or decode or golden" > $O/tests_parity.txt 2>&1
grep -E "passed|failed" $O/tests_parity.txt | tail -2
timeout 600 python tools/bench_decode.py > $O/decode.json 2>/dev/null; cat $O/decode.json
SH_GG_OUT3=0 timeout 600 python tools/bench_decode.py 2>/dev/null
timeout 300 python bench.py --steps 200 --warmup 20 --no-secondary --no-roofline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32', j['ms_per_step'])"
