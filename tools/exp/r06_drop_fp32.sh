#!/bin/bash
# round 6: training on the images (keep_fp32 == 2: fp32 rows neither pass reads are not written) - the check, then step time A/B (same box, alternating)
O=gpurun_out/r06drop; rm -rf $O; mkdir -p $O
timeout 900 python tools/drop_fp32_check.py > $O/check_drop1.txt 2>&1; tail -3 $O/check_drop1.txt | cut -c1-300
SH_P3_DROP_FP32=0 timeout 900 python tools/drop_fp32_check.py > $O/check_drop0.txt 2>&1; tail -3 $O/check_drop0.txt | cut -c1-300
cmp <(grep DIGEST $O/check_drop1.txt) <(grep DIGEST $O/check_drop0.txt) && echo "DIGESTS EQUAL"
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz f32 > $O/layer_drop1.txt 2>&1; grep -E "f32=0|total" $O/layer_drop1.txt | cut -c1-170
SH_F32_MMA=planes3 SH_P3_DROP_FP32=0 timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz f32 > $O/layer_drop0.txt 2>&1; grep -E "total" $O/layer_drop0.txt
for rep in 1 2; do for cfg in 1 0; do
  SH_P3_DROP_FP32=$cfg timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_drop${cfg}_$rep.json 2>$O/bench_drop${cfg}_$rep.err
  echo "== f32 drop=$cfg rep $rep: $(python -c "import json; d=json.loads([l for l in open('$O/bench_drop${cfg}_$rep.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)"
done; done
for cfg in 1 0; do
  SH_P3_DROP_FP32=$cfg timeout 300 python bench.py --steps 40 --warmup 10 --batch 32 --template tests/golden/template27554.npz --no-cpu-baseline --no-secondary --no-roofline > $O/bench_c4_drop$cfg.json 2>$O/bench_c4_drop$cfg.err
  echo "== config 4 drop=$cfg: $(python -c "import json; d=json.loads([l for l in open('$O/bench_c4_drop$cfg.json') if l.startswith('{')][-1]); print(d['ms_per_step'])" 2>&1 | tail -1)"
done
