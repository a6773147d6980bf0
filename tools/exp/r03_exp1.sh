#!/bin/bash
# round 3, experiment 1: bf16x3 form of the direct conv kernel - parity under the unchanged tests + per-launch times
O=gpurun_out/r03e1; mkdir -p $O
timeout 300 python tools/layer_report.py 64 > $O/lr_exact.txt 2>$O/lr_exact.err
for cfg in "0 4" "1 4" "2 4" "1 2" "2 2" "1 8"; do
  set -- $cfg
  SH_F32_MMA=split3 SH_S3_RT=$1 SH_S3_NT=$2 timeout 300 python tools/layer_report.py 64 > $O/lr_s3_rt$1_nt$2.txt 2>$O/lr_s3_rt$1_nt$2.err
done
SH_F32_MMA=split3 timeout 900 python -m pytest tests -q -m gpu -x -q -k "not bf16" > $O/tests_s3.txt 2>&1
SH_F32_MMA=split3 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu > $O/tests_s3_parity_all.txt 2>&1
tail -3 $O/tests_s3.txt $O/tests_s3_parity_all.txt
grep -h "total library" $O/lr_*.txt
