#!/bin/bash
O=gpurun_out/r05e17; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
SH_P3S_FORM=1 timeout 900 python -m pytest tests/test_p3.py tests/test_headline.py tests/test_configs.py -q -m gpu -x > $O/tests.txt 2>&1; tail -2 $O/tests.txt | cut -c1-200
for rep in 1 2; do for f in 0 1; do
  SH_P3S_FORM=$f SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/layer_f${f}_$rep.txt 2>&1
  echo "--- form $f rep $rep"; grep -h "conv_p3s\|total" $O/layer_f${f}_$rep.txt
done; done
for f in 0 1; do
  SH_P3S_FORM=$f SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 32 tests/golden/template27554.npz f32 > $O/layer_c4_f$f.txt 2>&1
  echo "--- c4 form $f"; grep -h "conv_p3s\|total" $O/layer_c4_f$f.txt
done
