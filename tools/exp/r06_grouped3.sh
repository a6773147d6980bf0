#!/bin/bash
# round 6: GPU suite on the grouped build; decode (config 5) with grouped lists, and the level-0 layer on the grouped plane kernel at large batch
O=gpurun_out/r06grp3; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
for cfg in "SH_P3_GROUPED=1" "SH_P3_GROUPED=0" "SH_P3_N16_MAXB=4096" "SH_P3_GROUPED=1" "SH_P3_GROUPED=0" "SH_P3_N16_MAXB=4096"; do
  env SH_F32_MMA=planes3 $cfg timeout 300 python tools/bench_decode.py --latents 40960 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('decode $cfg: p50 %.4f ms mean %.4f' % (d['p50_batch_ms'], d['mean_batch_ms']))"
done
SH_F32_MMA=planes3 timeout 300 python tools/layer_report_decode.py 2>/dev/null | cut -c1-150 > $O/layer_decode_grp1.txt; cat $O/layer_decode_grp1.txt
SH_F32_MMA=planes3 SH_P3_N16_MAXB=4096 timeout 300 python tools/layer_report_decode.py 2>/dev/null | cut -c1-150 > $O/layer_decode_n16.txt; grep -E "R=6891|total" $O/layer_decode_n16.txt
