#!/bin/bash
# round 3, experiment 3: coalesced-gather conv kernel (exact fp32 MFMA): parity (bit-identical to the direct form) + per-launch times
O=gpurun_out/r03e3; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ta_probe tools/exp/ta_probe.hip 2>/dev/null && timeout 120 /tmp/ta_probe > $O/ta_probe.txt 2>&1
cat $O/ta_probe.txt
for cfg in "0 4" "1 4" "2 4" "1 2" "2 2" "1 8" "2 8"; do
  set -- $cfg
  SH_GG_CG=1 SH_CG_RT=$1 SH_CG_NT=$2 timeout 300 python tools/layer_report.py 64 > $O/lr_cg_rt$1_nt$2.txt 2>$O/lr_cg_rt$1_nt$2.err
done
SH_GG_FILL=1 timeout 300 python tools/layer_report.py 64 > $O/lr_exact_fill1.txt 2>&1
SH_GG_CG=1 timeout 900 python -m pytest tests -q -m gpu -x -k "not bf16" > $O/tests_cg.txt 2>&1
tail -n 3 $O/tests_cg.txt
grep -h "total library" $O/lr_*.txt
