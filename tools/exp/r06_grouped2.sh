#!/bin/bash
# round 6: grouped lists with the "enough items" rule; the re-stated bf16 matched-L2 test under both backward-data forms
O=gpurun_out/r06grp2; rm -rf $O; mkdir -p $O
for cfg in 1 0; do SH_BF16_RAGGED=$cfg timeout 600 python -m pytest tests/test_bf16.py -m gpu -q -k "matched_l2" 2>&1 | grep -E "AssertionError: |passed|failed" | head -3; done
timeout 900 python -m pytest tests/test_p3.py tests/test_headline.py tests/test_bf16.py -m gpu -x -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz f32 > $O/layer_grp1.txt 2>&1; grep -E "conv_p3|total" $O/layer_grp1.txt | cut -c1-170
for rep in 1 2; do for cfg in 1 0; do
  SH_P3_GROUPED=$cfg timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_grp${cfg}_$rep.json 2>$O/bench_grp${cfg}_$rep.err
  echo "== f32 grouped=$cfg rep $rep: $(python -c "import json; d=json.loads([l for l in open('$O/bench_grp${cfg}_$rep.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)"
done; done
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 32 tests/golden/template27554.npz f32 > $O/layer_c4_grp1.txt 2>&1; grep -E "conv_p3|total" $O/layer_c4_grp1.txt | cut -c1-170
SH_P3_GROUPED=0 SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 32 tests/golden/template27554.npz f32 > $O/layer_c4_grp0.txt 2>&1; grep -E "conv_p3|total" $O/layer_c4_grp0.txt | cut -c1-170
for cfg in 1 0; do
  SH_P3_GROUPED=$cfg timeout 400 python bench.py --steps 40 --warmup 10 --batch 32 --template tests/golden/template27554.npz --no-cpu-baseline --no-secondary --no-roofline > $O/bench_c4_grp$cfg.json 2>$O/bench_c4_grp$cfg.err
  echo "== config 4 grouped=$cfg: $(python -c "import json; d=json.loads([l for l in open('$O/bench_c4_grp$cfg.json') if l.startswith('{')][-1]); print(d['ms_per_step'])" 2>&1 | tail -1)"
done
timeout 300 python tools/bench_decode.py --latents 20480 2>&1 | tail -2 | cut -c1-400
SH_P3_GROUPED=0 timeout 300 python tools/bench_decode.py --latents 20480 2>&1 | tail -2 | cut -c1-400
