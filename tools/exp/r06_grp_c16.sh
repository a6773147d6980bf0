#!/bin/bash
# round 6: grouped lists for the layers that gather 16 channels (conv_p3g_kernel<.., C16>): float64 gate, per-layer time, step A/B
O=gpurun_out/r06c16; rm -rf $O; mkdir -p $O
timeout 600 python tools/p3_probe.py 64 --both > $O/p3_probe.txt 2>&1; grep -E "K=160|K=176|rror" $O/p3_probe.txt | cut -c1-330
timeout 900 python -m pytest tests/test_p3.py tests/test_headline.py -m gpu -x -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz f32 > $O/layer_c16_1.txt 2>&1; grep -E "K=160|K=320|K=176|total" $O/layer_c16_1.txt | cut -c1-170
SH_P3_GRP_C16=0 SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz f32 > $O/layer_c16_0.txt 2>&1; grep -E "K=160|K=320|K=176|total" $O/layer_c16_0.txt | cut -c1-170
for rep in 1 2; do for cfg in 1 0; do
  SH_P3_GRP_C16=$cfg timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-roofline > $O/bench.json 2>$O/bench.err
  echo "== f32 c16 grouped=$cfg rep $rep: $(python -c "import json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)"
done; done
for cfg in 1 0; do
  SH_P3_GRP_C16=$cfg timeout 400 python bench.py --steps 40 --warmup 10 --batch 32 --template tests/golden/template27554.npz --no-cpu-baseline --no-secondary --no-roofline > $O/bench_c4.json 2>$O/bench_c4.err
  echo "== config 4 c16 grouped=$cfg: $(python -c "import json; d=json.loads([l for l in open('$O/bench_c4.json') if l.startswith('{')][-1]); print(d['ms_per_step'])" 2>&1 | tail -1)"
  SH_P3_GRP_C16=$cfg SH_F32_MMA=planes3 timeout 300 python tools/bench_decode.py --latents 20480 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('decode c16 grouped=$cfg: p50 %.4f ms' % d['p50_batch_ms'])"
done
