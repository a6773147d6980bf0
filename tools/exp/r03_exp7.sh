#!/bin/bash
# round 3, experiment 7: latent FCs in the bf16x3 form; every fp32 GPU parity test in both arithmetic forms; bench as the driver runs it
O=gpurun_out/r03e7; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu > $O/tests_all.txt 2>&1
grep -n "passed\|failed\|FAILED" $O/tests_all.txt | tail -12
SH_F32_MMA=split3 timeout 300 python tools/layer_report.py 64 > $O/lr_s3.txt 2>$O/lr_s3.err
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2>$O/bench_full.err
tail -c 400 $O/bench_full.err
grep -h "total library\|linear\|tgemm\|reduce" $O/lr_s3.txt
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r03e7/bench_full.json").read().splitlines() if l.startswith("{")][-1])
print(d["ms_per_step"], d["config"]["workload"][:80], d["roofline"]["kernel"], d["roofline"]["frac"], d["matched_l2"]["rel_diff"])
print({k:(v.get("ms_per_step") or v.get("ms_per_iteration") or v.get("p50_batch_ms") or v.get("error")) for k,v in d.get("secondary",{}).items()})
PY
