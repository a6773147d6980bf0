#!/bin/bash
O=gpurun_out/r05e19; rm -rf $O; mkdir -p $O
for it in 1024 1536 2048 768; do
  SH_WS_ITEMS=$it SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/items_$it.txt 2>&1
  echo "--- SH_WS_ITEMS=$it"; grep -h "wgrad_stream\|total\|slab_reduce" $O/items_$it.txt | awk '{printf "%s ", $(NF-1)} /wgrad_stream/ {s+=$(NF-1)} END {print " | wgrad_stream sum", s}'
done
for d in 16 64; do
  SH_WS_SLAB_MB=$d SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/slab_$d.txt 2>&1
  echo "--- SH_WS_SLAB_MB=$d"; grep -h "wgrad_stream\|total\|slab_reduce" $O/slab_$d.txt | awk '{printf "%s ", $(NF-1)} /wgrad_stream/ {s+=$(NF-1)} END {print " | wgrad_stream sum", s}'
done
