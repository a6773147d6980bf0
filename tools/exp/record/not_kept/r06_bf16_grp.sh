#!/bin/bash
# round 6: grouped lists on the bf16 path (conv_bf16g_kernel): unit tests, layer report, step A/B (same box, alternating)
O=gpurun_out/r06bfg; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_bf16.py -m gpu -x -q -k "conv_fwd_and_bwd" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz bf16 > $O/layer_grp1.txt 2>&1; grep -E "conv_bf16|total" $O/layer_grp1.txt | cut -c1-160
SH_BF16_GROUPED=0 timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz bf16 > $O/layer_grp0.txt 2>&1; grep -E "conv_bf16|total" $O/layer_grp0.txt | cut -c1-160
for rep in 1 2; do for cfg in 1 0; do
  SH_BF16_GROUPED=$cfg timeout 300 python bench.py --steps 100 --warmup 10 --dtype bf16 --no-cpu-baseline --no-secondary --no-roofline > $O/bench.json 2>$O/bench.err
  echo "== bf16 grouped=$cfg rep $rep: $(python -c "import json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)"
done; done
for cfg in 1 0; do
  SH_BF16_GROUPED=$cfg timeout 400 python bench.py --steps 40 --warmup 10 --dtype bf16 --batch 32 --template tests/golden/template27554.npz --no-cpu-baseline --no-secondary --no-roofline > $O/bench_c4.json 2>$O/bench_c4.err
  echo "== config 4 bf16 grouped=$cfg: $(python -c "import json; d=json.loads([l for l in open('$O/bench_c4.json') if l.startswith('{')][-1]); print(d['ms_per_step'])" 2>&1 | tail -1)"
done
timeout 900 python -m pytest tests/test_bf16.py -m gpu -x -q > $O/tests_all.txt 2>&1; tail -3 $O/tests_all.txt
