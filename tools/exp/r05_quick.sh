#!/bin/bash
cd "$(dirname "$0")/../.."
rep() { SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 2>/dev/null | grep -E "^gather_gemm<1|conv_out3|wgrad_thin|total" | awk '{printf "%s %s | ", $1, $(NF-1)} END {print ""}'; }
echo "default: $(rep)"
for v in 4 8 32 64; do echo "SH_GG_TB=$v: $(SH_GG_TB=$v rep)"; done
for v in 256 512 1536 3072; do echo "SH_GG_FILL=$v: $(SH_GG_FILL=$v rep)"; done
for v in 1 2; do echo "SH_GG_RT=$v: $(SH_GG_RT=$v rep)"; done
for v in 32 128 256; do echo "SH_OUT3_WG_PER_XCD=$v: $(SH_OUT3_WG_PER_XCD=$v rep)"; done
echo "default: $(rep)"
