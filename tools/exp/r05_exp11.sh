#!/bin/bash
O=gpurun_out/r05e11; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1200 python -m pytest tests/test_p3.py tests/test_gpu_parity.py tests/test_headline.py -q -m gpu -x > $O/tests.txt 2>&1; tail -3 $O/tests.txt | cut -c1-300
for v in 1 0; do
  SH_SPMM_P3X8=$v SH_F32_MMA=planes3 timeout 300 python tools/layer_report_decode.py > $O/decode_x8_$v.txt 2>&1
  SH_SPMM_P3X8=$v SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/layer_x8_$v.txt 2>&1
  echo "--- x8=$v"; grep -h "spmm\|total" $O/decode_x8_$v.txt $O/layer_x8_$v.txt
done
