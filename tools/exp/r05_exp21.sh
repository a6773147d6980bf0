#!/bin/bash
O=gpurun_out/r05e21; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_configs.py tests/test_headline.py tests/test_p3.py -q -m gpu -x > $O/tests.txt 2>&1; tail -2 $O/tests.txt | cut -c1-200
for v in 256 1000000; do
  SH_P3_N16_MAXB=$v SH_F32_MMA=planes3 timeout 300 python tools/layer_report_decode.py > $O/decode_$v.txt 2>&1
  echo "--- SH_P3_N16_MAXB=$v"; cat $O/decode_$v.txt | tail -13
done
SH_F32_MMA=planes3 timeout 300 python tools/bench_decode.py --latents 20480 > $O/decode.json 2>&1; tail -1 $O/decode.json | cut -c1-400
