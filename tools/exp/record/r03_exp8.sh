#!/bin/bash
# round 3, experiment 8: bf16x3 conv with line-wise gathers (s3c): tile variants, parity in both forms, step time
O=gpurun_out/r03e8; mkdir -p $O
export SH_F32_MMA=split3
for cfg in "0 4 4" "1 8 4" "0 8 4" "0 4 2" "2 4 4"; do
  set -- $cfg
  SH_S3_RT=$1 SH_S3_NT=$2 SH_S3_MIN_NT=$3 timeout 300 python tools/layer_report.py 64 > $O/lr_s3c_rt$1_nt$2_min$3.txt 2>$O/lr_s3c_rt$1_nt$2_min$3.err
done
SH_S3_CO=0 timeout 300 python tools/layer_report.py 64 > $O/lr_s3_noco.txt 2>&1
unset SH_F32_MMA
timeout 1200 python -m pytest tests -q -m gpu > $O/tests_all.txt 2>&1
grep -n "passed\|failed\|FAILED" $O/tests_all.txt | tail -12
timeout 300 python bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench_s3c.json 2>$O/bench_s3c.err
grep -h "total library" $O/lr_*.txt
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r03e8/bench_s3c.json").read().splitlines() if l.startswith("{")][-1])
print(d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("frac_of_f32_mfma_peak"))
PY
