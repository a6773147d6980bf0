#!/bin/bash
# round 6: what bounds conv_p3_kernel - the same launches with parts of the k-step removed (diagnostic libraries built with -DP3_ABLATE=n:
# for a in 1 2 3 4; do hipcc ... -DP3_ABLATE=$a -c p3_conv.hip; link with the other objects into semantichuman_amd/lib_abl$a/; done)
# 0 = the shipped kernel, 1 = no gathered loads, 2 = the loads alone, 3 = loads + weight-fragment reads from LDS (no MFMAs), 4 = loads + MFMAs (no LDS reads)
O=gpurun_out/r06abl; rm -rf $O; mkdir -p $O
for a in 0 1 2 3 4; do
  lib=""; [ $a != 0 ] && lib=$PWD/semantichuman_amd/lib_abl$a/libsh_kernels.so
  SH_KERNEL_LIB=$lib SH_P3_GROUPED=0 timeout 600 python tools/p3_probe.py 64 --both --reps=20 > $O/probe_$a.txt 2>&1
  echo "== ablate $a"; grep -E "fwd|bwd" $O/probe_$a.txt | awk '{printf "%s %s %s %s  p3 %s us\n", $1, $2, $3, $4, $11}'
done
