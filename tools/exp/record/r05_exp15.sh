#!/bin/bash
O=gpurun_out/r05e15; rm -rf $O; mkdir -p $O
for rep in 1 2; do
  SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/layer_base_$rep.txt 2>&1
  SH_KERNEL_LIB=$PWD/semantichuman_amd/lib_alt/libsh_kernels.so SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/layer_alt_$rep.txt 2>&1
  for f in base alt; do echo "--- $f $rep"; grep -h "wgrad_stream\|total" $O/layer_${f}_$rep.txt | awk '{print $(NF-1)} /wgrad_stream/ {s+=$(NF-1)} END {print "wgrad_stream sum", s}' | paste -sd' '; done
done
