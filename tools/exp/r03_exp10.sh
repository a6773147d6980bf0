#!/bin/bash
# round 3, experiment 10: weight gradients in the bf16x3 form
O=gpurun_out/r03e10; mkdir -p $O
SH_F32_MMA=split3 timeout 300 python tools/layer_report.py 64 > $O/lr_s3.txt 2>$O/lr_s3.err
SH_F32_MMA=split3 SH_S3_WG_MIN_COT=4 timeout 300 python tools/layer_report.py 64 > $O/lr_s3_cot4.txt 2>$O/lr_s3_cot4.err
grep -h "wgrad\|total" $O/lr_s3.txt | cut -c1-110
grep -h "wgrad\|total" $O/lr_s3_cot4.txt | cut -c1-110
timeout 1200 python -m pytest tests -q -m gpu > $O/tests_all.txt 2>&1
grep -n "passed\|failed\|FAILED" $O/tests_all.txt | tail -12
timeout 300 python bench.py --steps 30 --warmup 5 --no-secondary > $O/bench_s3.json 2>$O/bench_s3.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r03e10/bench_s3.json").read().splitlines() if l.startswith("{")][-1])
print(d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["matched_l2"])
PY
