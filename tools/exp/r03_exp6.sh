#!/bin/bash
# round 3, experiment 6: bf16x3 form on the layers with >= 4 channel tiles: parity under the unchanged tests (6 and 9 products), step time
O=gpurun_out/r03e6; mkdir -p $O
SH_F32_MMA=split3 timeout 900 python -m pytest tests -q -m gpu -k "not bf16" > $O/tests_s3.txt 2>&1
grep -n "passed\|failed\|FAILED" $O/tests_s3.txt | tail -8
SH_F32_MMA=split3 SH_S3_ALL9=1 timeout 900 python -m pytest tests -q -m gpu -k "not bf16" > $O/tests_s3_all9.txt 2>&1
grep -n "passed\|failed\|FAILED" $O/tests_s3_all9.txt | tail -8
SH_F32_MMA=split3 timeout 300 python tools/layer_report.py 64 > $O/lr_s3.txt 2>$O/lr_s3.err
SH_F32_MMA=split3 SH_S3_ALL9=1 timeout 300 python tools/layer_report.py 64 > $O/lr_s3_all9.txt 2>$O/lr_s3_all9.err
SH_F32_MMA=split3 timeout 300 python bench.py --steps 30 --warmup 5 --no-secondary --no-roofline > $O/bench_s3.json 2>$O/bench_s3.err
SH_F32_MMA=split3 SH_S3_ALL9=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-secondary --no-roofline > $O/bench_s3_all9.json 2>$O/bench_s3_all9.err
grep -h "total library" $O/lr_*.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03e6/bench_*.json")):
    try:
        d=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1]); print(f, round(d["ms_per_step"],4), d.get("matched_l2"))
    except Exception as e: print(f,"ERR",e)
PY
