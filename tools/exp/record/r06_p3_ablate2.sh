#!/bin/bash
# round 6: what is left of conv_p3_kernel WITHOUT the gather (item 13 found 75 % of a two-tile launch there) - the epilogue's share.
# Diagnostic libraries built out of tree from a patched copy of csrc/p3_conv.hip (-DP3_ABLATE=n, semantichuman_amd/lib_abl$n/):
# 5 = no epilogue stores, 6 = no gathered loads and no stores (weight fill + LDS / MFMA chain alone), 7 = the loads alone without stores,
# 8 = no fp32 store (image only), 9 = no image store (fp32 only); 0 = the shipped kernel
O=gpurun_out/r06abl2; rm -rf $O; mkdir -p $O
for a in 0 5 6 7 8 9; do
  lib=""; [ $a != 0 ] && lib=$PWD/semantichuman_amd/lib_abl$a/libsh_kernels.so
  SH_KERNEL_LIB=$lib SH_P3_GROUPED=0 timeout 600 python tools/p3_probe.py 64 --both --reps=20 > $O/probe_$a.txt 2>&1
  echo "== ablate $a: $(grep -E 'fwd|bwd' $O/probe_$a.txt | awk '{printf "%s%s=%s ", $1, $2, $11}')"
done
