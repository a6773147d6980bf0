#!/bin/bash
O=gpurun_out/r03e9; mkdir -p $O
export SH_F32_MMA=split3
for cfg in "1 4" "2 4" "2 8" "1 8"; do
  set -- $cfg
  SH_S3_RT=$1 SH_S3_NT=$2 timeout 300 python tools/layer_report.py 64 > $O/lr_s3_rt$1_nt$2.txt 2>$O/lr_s3_rt$1_nt$2.err
  grep -h "split3" $O/lr_s3_rt$1_nt$2.txt | cut -c1-110
  grep -h "total library" $O/lr_s3_rt$1_nt$2.txt
done
