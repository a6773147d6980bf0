#!/bin/bash
O=gpurun_out/r05e20; rm -rf $O; mkdir -p $O
for b in 128 256 512 1024; do
  SH_F32_MMA=exact timeout 300 python tools/layer_report_decode.py $b > $O/decode_exact_$b.txt 2>&1
  echo "--- exact B=$b (per 64 meshes)"; grep -h "gather_gemm\|spmm\|total" $O/decode_exact_$b.txt | awk -v b=$b '{v=$(NF-1)+0; if (v==0) v=$NF+0; printf "%s %s %s  %.1f\n", $1, $2, $3, v*64/b}' | grep "R=6891\|rows=3445\|total"
done
