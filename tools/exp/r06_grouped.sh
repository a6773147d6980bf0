#!/bin/bash
# round 6: plane convs over GROUPED lists (conv_p3g_kernel) - float64 gate, per-layer times, step time A/B (same box, alternating)
O=gpurun_out/r06grp; rm -rf $O; mkdir -p $O
timeout 600 python tools/p3_probe.py 64 --both > $O/p3_probe.txt 2>&1; grep -E "grouped|Error|error" $O/p3_probe.txt | cut -c1-330
timeout 900 python -m pytest tests/test_p3.py tests/test_headline.py -m gpu -x -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz f32 > $O/layer_grp1.txt 2>&1; grep -E "conv_p3|total" $O/layer_grp1.txt | cut -c1-170
SH_F32_MMA=planes3 SH_P3_GROUPED=0 timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz f32 > $O/layer_grp0.txt 2>&1; grep -E "total" $O/layer_grp0.txt
for rep in 1 2; do for cfg in 1 0; do
  SH_P3_GROUPED=$cfg timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_grp${cfg}_$rep.json 2>$O/bench_grp${cfg}_$rep.err
  echo "== f32 grouped=$cfg rep $rep: $(python -c "import json; d=json.loads([l for l in open('$O/bench_grp${cfg}_$rep.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)"
done; done
for cfg in 1 0; do
  SH_P3_GROUPED=$cfg timeout 400 python bench.py --steps 40 --warmup 10 --batch 32 --template tests/golden/template27554.npz --no-cpu-baseline --no-secondary --no-roofline > $O/bench_c4_grp$cfg.json 2>$O/bench_c4_grp$cfg.err
  echo "== config 4 grouped=$cfg: $(python -c "import json; d=json.loads([l for l in open('$O/bench_c4_grp$cfg.json') if l.startswith('{')][-1]); print(d['ms_per_step'])" 2>&1 | tail -1)"
done
bash tools/exp/r06_bf16_rag_err.sh
