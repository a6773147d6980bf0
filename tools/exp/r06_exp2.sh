#!/bin/bash
# round 6, experiment 2: config 4 (27 554 vertices, batch 32) with the three-plane weight gradient; the sharded update in a world of one
O=gpurun_out/r06e2; rm -rf $O; mkdir -p $O
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 32 tests/golden/template27554.npz f32 > $O/layer_config4_p3w.txt 2>&1; grep -E "wgrad|total|spmm" $O/layer_config4_p3w.txt
SH_P3_WGRAD=0 SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 32 tests/golden/template27554.npz f32 > $O/layer_config4_exactw.txt 2>&1; grep -E "wgrad|total" $O/layer_config4_exactw.txt
for cfg in "base" "SH_P3_WGRAD=0"; do
  if [ "$cfg" = "base" ]; then timeout 300 python bench.py --steps 40 --warmup 10 --batch 32 --template tests/golden/template27554.npz --no-cpu-baseline --no-secondary --no-roofline > $O/bench_c4_$cfg.json 2>$O/bench_c4_$cfg.err
  else env $cfg timeout 300 python bench.py --steps 40 --warmup 10 --batch 32 --template tests/golden/template27554.npz --no-cpu-baseline --no-secondary --no-roofline > "$O/bench_c4_$cfg.json" 2>"$O/bench_c4_$cfg.err"; fi
  echo "== config4 $cfg: $(python -c "import json,sys; d=json.loads([l for l in open('$O/bench_c4_$cfg.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)"
done
SH_BENCH_FORCE_REDUCER=1 timeout 600 python bench.py --steps 20 --warmup 5 --dtype bf16 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_reducer_bf16_sharded.json 2>$O/bench_reducer_bf16_sharded.err
python -c "import json; d=json.loads([l for l in open('$O/bench_reducer_bf16_sharded.json') if l.startswith('{')][-1]); print('reducer bf16 (sharded default):', d['ms_per_step'], d['config'].get('launch'), d['config'].get('parallelism'))" 2>&1 | tail -2
tail -3 $O/bench_reducer_bf16_sharded.err
