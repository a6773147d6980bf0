#!/bin/bash
cd "$(dirname "$0")/../.."
for lib in "" alt/libsh_kernels_d3w6.so alt/libsh_kernels_d2w6.so alt/libsh_kernels_d2w8.so; do
  echo "=== lib=${lib:-default}"
  SH_KERNEL_LIB=${lib:+$PWD/semantichuman_amd/lib/$lib} SH_P3_RT=1 python tools/p3_probe.py 64 --both --reps=10 2>&1 | grep -E "fwd|bwd" | awk '{print $1,$2,$3,$4,$5, $7, $8, $11, $12, $13, $14}'
done
