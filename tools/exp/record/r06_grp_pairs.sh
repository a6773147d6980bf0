#!/bin/bash
# round 6: two-row groups for the launches that have too few four-row groups at the batch (SH_P3_GROUPED_PAIRS): per-layer times, step A/B
O=gpurun_out/r06pairs; rm -rf $O; mkdir -p $O
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz f32 > $O/layer_1.txt 2>&1; grep -E "conv_p3|total" $O/layer_1.txt | cut -c1-170
SH_P3_GROUPED_PAIRS=0 SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz f32 > $O/layer_0.txt 2>&1; grep -E "conv_p3|total" $O/layer_0.txt | cut -c1-170
for rep in 1 2; do for cfg in 1 0; do
  SH_P3_GROUPED_PAIRS=$cfg timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-roofline > $O/bench.json 2>$O/bench.err
  echo "== f32 pairs=$cfg rep $rep: $(python -c "import json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)"
done; done
for cfg in 1 0; do
  SH_P3_GROUPED_PAIRS=$cfg timeout 400 python bench.py --steps 40 --warmup 10 --batch 32 --template tests/golden/template27554.npz --no-cpu-baseline --no-secondary --no-roofline > $O/bench_c4.json 2>$O/bench_c4.err
  echo "== config 4 pairs=$cfg: $(python -c "import json; d=json.loads([l for l in open('$O/bench_c4.json') if l.startswith('{')][-1]); print(d['ms_per_step'])" 2>&1 | tail -1)"
done
