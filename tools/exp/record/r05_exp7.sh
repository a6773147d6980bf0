#!/bin/bash
O=gpurun_out/r05e7; rm -rf $O; mkdir -p $O
for k in 0 3 2; do
  SH_P3_PROBE_SKIP=$k SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/layer_skip$k.txt 2>&1
  echo "--- skip $k"; grep -h "conv_p3<\|total" $O/layer_skip$k.txt
done
