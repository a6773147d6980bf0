#!/bin/bash
# line-wise VALU forward of the 16 -> 3 layer: parity + A/B
set -u
O=gpurun_out/r03e19; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv_kernels_vs_oracle or launch_table or golden or stack" > $O/tests_parity.txt 2>&1
tail -3 $O/tests_parity.txt
for g in 1 0; do
  SH_GG_OUT3=$g timeout 300 python bench.py --steps 200 --warmup 20 --no-secondary > $O/bench_out3_$g.json 2> $O/bench_out3_$g.err
done
for w in 32 96 128; do
  SH_OUT3_WG_PER_XCD=$w timeout 300 python bench.py --steps 200 --warmup 20 --no-secondary --no-roofline > $O/bench_wg_$w.json 2> $O/bench_wg_$w.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03e19/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], j["ms_per_step"], j["build"]["env"])
    except Exception as e: print(f, "ERR", e)
PY
python tools/layer_report.py > $O/layer_report.txt 2>&1 || true
grep -E "out3|N=3 " $O/layer_report.txt; tail -1 $O/layer_report.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests_all.txt 2>&1; tail -3 $O/tests_all.txt
