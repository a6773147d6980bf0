#!/bin/bash
O=gpurun_out/r05e25; rm -rf $O; mkdir -p $O
for rep in 1 2; do
  SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/base_$rep.txt 2>&1
  for v in 0 3; do SH_KERNEL_LIB=$PWD/semantichuman_amd/lib_alt$v/libsh_kernels.so SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/alt${v}_$rep.txt 2>&1; done
  for f in base alt0 alt3; do echo "--- $f $rep"; grep -h "wgrad_stream" $O/${f}_$rep.txt | awk '{printf "%s ", $(NF-1)} {s+=$(NF-1)} END {print " | sum", s}'; done
done
