#!/bin/bash
# round 6: grouped lists, the "enough items" rule against every layer grouped (written for a balancing pass of group_lists - unions capped at 1.15 x
# their mean - that was measured and dropped: item 14 of the record; what it re-runs today is the comparison of the rule): per-layer times standalone, in the step with the
# "enough items" rule and with every layer grouped, step time A/B
O=gpurun_out/r06grp4; rm -rf $O; mkdir -p $O
timeout 600 python tools/p3_probe.py 64 --both > $O/p3_probe.txt 2>&1; grep -E "grouped" $O/p3_probe.txt | awk '{print $1,$2,$3,$4,"p3",$11; for(i=1;i<=NF;i++) if($i=="grouped"||$i=="lists") printf "   %s ... %s %s\n", $i, $(NF-1), $NF}' | cut -c1-200
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz f32 > $O/layer_rule.txt 2>&1; grep -E "conv_p3g|conv_p3r|conv_p3<|total" $O/layer_rule.txt | cut -c1-170
SH_P3_GRP_MIN_ITEMS_PER_CU=0 SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz f32 > $O/layer_all.txt 2>&1; grep -E "conv_p3g|total" $O/layer_all.txt | cut -c1-170
for rep in 1 2; do for cfg in "SH_P3_GROUPED=1" "SH_P3_GROUPED=0" "SH_P3_GRP_MIN_ITEMS_PER_CU=0"; do
  env $cfg timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-roofline > $O/bench.json 2>$O/bench.err
  echo "== f32 $cfg rep $rep: $(python -c "import json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)"
done; done
for cfg in "SH_P3_GROUPED=1" "SH_P3_GROUPED=0"; do
  env $cfg timeout 400 python bench.py --steps 40 --warmup 10 --batch 32 --template tests/golden/template27554.npz --no-cpu-baseline --no-secondary --no-roofline > $O/bench_c4.json 2>$O/bench_c4.err
  echo "== config 4 $cfg: $(python -c "import json; d=json.loads([l for l in open('$O/bench_c4.json') if l.startswith('{')][-1]); print(d['ms_per_step'])" 2>&1 | tail -1)"
done
