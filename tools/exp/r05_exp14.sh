#!/bin/bash
O=gpurun_out/r05e14; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_headline.py tests/test_configs.py -q -m gpu -x > $O/tests.txt 2>&1; tail -3 $O/tests.txt | cut -c1-250
for rep in 1 2; do
  SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/layer_$rep.txt 2>&1
  echo "--- rep $rep"; grep -h "wgrad_stream\|total" $O/layer_$rep.txt | awk '{print $0} /wgrad_stream/ {s+=$(NF-1)} END {print "wgrad_stream sum", s}'
done
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 32 tests/golden/template27554.npz f32 > $O/layer_c4.txt 2>&1
echo "--- c4"; grep -h "wgrad_stream\|total" $O/layer_c4.txt | awk '{print $0} /wgrad_stream/ {s+=$(NF-1)} END {print "wgrad_stream sum", s}'
SH_F32_MMA=exact timeout 300 python tools/layer_report.py 64 > $O/layer_exact.txt 2>&1; grep -h "total" $O/layer_exact.txt
