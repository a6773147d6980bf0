#!/bin/bash
# Diagnostic: builds libsh_kernels variants with parts of the wgrad kernel removed (SH_WG_ABLATE bits:
# 1 no global loads, 2 no MFMA, 4 no LDS stores, 8 no LDS reads) and prints the per-launch wgrad times of each.
# Results of ablated builds are WRONG by construction - timing only.  Run on the GPU box.
cd "$(dirname "$0")/.."
for a in 0 1 2 4 8 3 12 6; do
    out=/tmp/libsh_ablate_$a.so
    make -s -C semantichuman_amd/csrc OUT=$out FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -DSH_WG_ABLATE=$a" >/dev/null 2>&1
    echo "== SH_WG_ABLATE=$a"
    SH_KERNEL_LIB=$out SH_OVERLAP_WGRAD=0 python tools/layer_report.py 2>/dev/null | grep -E "^wgrad"
done
