#!/bin/bash
# round 6, experiment 1: the three-plane weight gradient inside the training step - tests, layer report, A/B of the step
O=gpurun_out/r06e1; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_p3.py -x -q -k "three_plane_weight" > $O/test_p3.txt 2>&1; tail -3 $O/test_p3.txt
timeout 900 python -m pytest tests/test_headline.py -x -q -k "template6890" > $O/test_headline.txt 2>&1; tail -3 $O/test_headline.txt
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/layer_planes3.txt 2>&1; grep -E "wgrad|spmm|total|slab" $O/layer_planes3.txt
for cfg in "base" "SH_WP3_TAIL=0" "SH_P3_WGRAD=0"; do
  if [ "$cfg" = "base" ]; then timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_$cfg.json 2>$O/bench_$cfg.err
  else env $cfg timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-secondary --no-roofline > "$O/bench_$cfg.json" 2>"$O/bench_$cfg.err"; fi
  echo "== $cfg: $(python -c "import json,sys; d=json.loads([l for l in open('$O/bench_$cfg.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)"
done
