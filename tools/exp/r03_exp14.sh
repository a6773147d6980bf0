#!/bin/bash
# round 3, experiment 14: the last pre-sum level of a layer as tail workgroups of its weight-gradient launch
O=gpurun_out/r03e14; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/tests_all.txt 2>&1
grep -n "passed\|failed\|FAILED" $O/tests_all.txt | tail -8
timeout 300 python tools/layer_report.py 64 > $O/lr_tail.txt 2>$O/lr_tail.err
SH_WS_TAIL=0 timeout 300 python tools/layer_report.py 64 > $O/lr_notail.txt 2>$O/lr_notail.err
SH_WS_TAIL_BLOCKS=256 timeout 300 python tools/layer_report.py 64 > $O/lr_tail256.txt 2>/dev/null
grep -h "total library" $O/lr_*.txt
grep -h "wgrad_stream\|spmm" $O/lr_tail.txt | cut -c1-120
for i in 1 2; do
timeout 300 python bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline > $O/bench_tail_$i.json 2>/dev/null
SH_WS_TAIL=0 timeout 300 python bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline > $O/bench_notail_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03e14/bench_*.json")):
    d=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1]); print(f.split("/")[-1], round(d["ms_per_step"],4))
PY
