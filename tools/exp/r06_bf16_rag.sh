#!/bin/bash
# round 6, bf16 backward-data over ragged source lists (conv_bf16r_kernel): parity tests, layer report, step time A/B (same box, alternating)
O=gpurun_out/r06bf; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_bf16.py -m gpu -x -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt
timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz bf16 > $O/layer_bf16_rag.txt 2>&1; grep -E "conv_bf16|spmm_bf16|total" $O/layer_bf16_rag.txt | cut -c1-150
SH_BF16_RAGGED=0 timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz bf16 > $O/layer_bf16_dense.txt 2>&1; grep -E "true|total" $O/layer_bf16_dense.txt | cut -c1-150
for rep in 1 2; do for cfg in 1 0; do
  SH_BF16_RAGGED=$cfg timeout 300 python bench.py --steps 100 --warmup 10 --dtype bf16 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_rag${cfg}_$rep.json 2>$O/bench_rag${cfg}_$rep.err
  echo "== bf16 ragged=$cfg rep $rep: $(python -c "import json; d=json.loads([l for l in open('$O/bench_rag${cfg}_$rep.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)"
done; done
SH_BF16_RAGGED=1 timeout 300 python bench.py --steps 40 --warmup 10 --dtype bf16 --batch 32 --template tests/golden/template27554.npz --no-cpu-baseline --no-secondary --no-roofline > $O/bench_c4_bf16.json 2>$O/bench_c4_bf16.err
echo "== config 4 bf16 ragged: $(python -c "import json; d=json.loads([l for l in open('$O/bench_c4_bf16.json') if l.startswith('{')][-1]); print(d['ms_per_step'])" 2>&1 | tail -1)"
SH_BF16_RAGGED=0 timeout 300 python bench.py --steps 40 --warmup 10 --dtype bf16 --batch 32 --template tests/golden/template27554.npz --no-cpu-baseline --no-secondary --no-roofline > $O/bench_c4_bf16_dense.json 2>$O/bench_c4_bf16_dense.err
echo "== config 4 bf16 dense: $(python -c "import json; d=json.loads([l for l in open('$O/bench_c4_bf16_dense.json') if l.startswith('{')][-1]); print(d['ms_per_step'])" 2>&1 | tail -1)"
