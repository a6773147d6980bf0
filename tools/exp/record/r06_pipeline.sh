#!/bin/bash
# round 6: the grouped plane conv with the next item's first loads issued BEFORE the epilogue of the current one: float64 gate, per-layer
# times standalone, step / config 4 / decode (compare with the previous build's numbers of the same scripts; the kernel has no switch)
O=gpurun_out/r06pipe; rm -rf $O; mkdir -p $O
timeout 600 python tools/p3_probe.py 64 --both > $O/p3_probe.txt 2>&1; grep -E "grouped" $O/p3_probe.txt | awk '{n=split($0,a,"grouped"); print $1,$2,$3,$4,"p3",$11,"| grouped",a[2]}' | cut -c1-200
timeout 900 python -m pytest tests/test_p3.py tests/test_headline.py -m gpu -x -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz f32 > $O/layer.txt 2>&1; grep -E "conv_p3g|total" $O/layer.txt | cut -c1-170
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 32 tests/golden/template27554.npz f32 > $O/layer_c4.txt 2>&1; grep -E "conv_p3g|total" $O/layer_c4.txt | cut -c1-170
SH_F32_MMA=planes3 timeout 300 python tools/layer_report_decode.py 2>/dev/null | cut -c1-150 > $O/layer_decode.txt; grep -E "conv_p3g|total" $O/layer_decode.txt
for rep in 1 2; do
  timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-roofline > $O/bench.json 2>$O/bench.err
  echo "== f32 rep $rep: $(python -c "import json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)"
  SH_P3_GROUPED=0 timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-roofline > $O/bench.json 2>$O/bench.err
  echo "== f32 ungrouped rep $rep: $(python -c "import json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)"
done
timeout 400 python bench.py --steps 40 --warmup 10 --batch 32 --template tests/golden/template27554.npz --no-cpu-baseline --no-secondary --no-roofline > $O/bench_c4.json 2>$O/bench_c4.err
echo "== config 4: $(python -c "import json; d=json.loads([l for l in open('$O/bench_c4.json') if l.startswith('{')][-1]); print(d['ms_per_step'])" 2>&1 | tail -1)"
SH_P3_GROUPED=0 timeout 400 python bench.py --steps 40 --warmup 10 --batch 32 --template tests/golden/template27554.npz --no-cpu-baseline --no-secondary --no-roofline > $O/bench_c4.json 2>$O/bench_c4.err
echo "== config 4 ungrouped: $(python -c "import json; d=json.loads([l for l in open('$O/bench_c4.json') if l.startswith('{')][-1]); print(d['ms_per_step'])" 2>&1 | tail -1)"
for cfg in 1 0; do SH_P3_GROUPED=$cfg SH_F32_MMA=planes3 timeout 300 python tools/bench_decode.py --latents 20480 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('decode grouped=$cfg: p50 %.4f ms' % d['p50_batch_ms'])"; done
