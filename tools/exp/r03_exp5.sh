#!/bin/bash
# round 3, experiment 5: bf16 conv with line-wise loads; fp32 skip as template
O=gpurun_out/r03e5; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu > $O/tests_all.txt 2>&1
grep -n "passed\|failed" $O/tests_all.txt | tail -2
timeout 300 python tools/layer_report.py 64 > $O/lr_f32.txt 2>$O/lr_f32.err
for co in 0 1 2; do
  SH_BC_CO=$co timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz bf16 > $O/lr_bf16_co$co.txt 2>$O/lr_bf16_co$co.err
  SH_BC_CO=$co timeout 300 python bench.py --dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_bf16_co$co.json 2>$O/bench_bf16_co$co.err
done
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_f32.json 2>$O/bench_f32.err
grep -h "total library" $O/lr_*.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03e5/bench_*.json")):
    try:
        d=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1]); print(f, round(d["ms_per_step"],4))
    except Exception as e: print(f,"ERR",e)
PY
