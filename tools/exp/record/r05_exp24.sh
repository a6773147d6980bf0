#!/bin/bash
O=gpurun_out/r05e24; rm -rf $O; mkdir -p $O
ALT=$PWD/semantichuman_amd/lib_alt/libsh_kernels.so
for rep in 1 2; do
  SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/base_$rep.txt 2>&1
  SH_KERNEL_LIB=$ALT SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/alt_$rep.txt 2>&1
  for f in base alt; do echo "--- $f $rep"; grep -h "wgrad_stream" $O/${f}_$rep.txt | awk '{printf "%s ", $(NF-1)} {s+=$(NF-1)} END {print " | sum", s}'; done
done
