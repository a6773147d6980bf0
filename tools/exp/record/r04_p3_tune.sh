#!/bin/bash
# ring depth / rows-per-wave variants of the three-plane conv kernels (alt builds in semantichuman_amd/lib/alt)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for lib in "" alt/libsh_kernels_b96.so alt/libsh_kernels_b100.so; do
  for rt in 0 1; do
    echo "=== lib=${lib:-default} SH_P3_RT=$rt"
    SH_KERNEL_LIB=${lib:+$PWD/semantichuman_amd/lib/$lib} SH_P3_RT=$rt python tools/p3_probe.py 64 --both --reps=10 2>&1 | grep -E "fwd|bwd" | awk '{print $1,$2,$3,$4,$5, $7, $8, $11, $12, $13, $14}'
  done
done
echo "=== streaming RT=1"
SH_P3S_RT=1 python tools/p3_probe.py 64 --both --reps=10 2>&1 | grep -E "enc3|dec1 " | awk '{print $1,$2,$3,$4,$5, $7, $8, $11, $12, $13, $14}'
echo "=== adversarial NP=6"
python tools/p3_probe.py 64 --both --adversarial --reps=2 2>&1 | grep -E "fwd|bwd"
echo "=== adversarial NP=9"
SH_P3_NP=9 python tools/p3_probe.py 64 --both --adversarial --reps=10 2>&1 | grep -E "fwd|bwd"
