#!/bin/bash
O=gpurun_out/r05e4; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_headline.py tests/test_p3.py tests/test_configs.py -q -m gpu -s > $O/tests.txt 2>&1; tail -5 $O/tests.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "conv_kernels or sequencing or switched" > $O/tests2.txt 2>&1; tail -3 $O/tests2.txt
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 32 tests/golden/template27554.npz f32 > $O/layer_c4_new.txt 2>&1
SH_P3S_PAD=0 SH_P3S_NT2=0 SH_P3_MAX_SPLIT=2 SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 32 tests/golden/template27554.npz f32 > $O/layer_c4_split2.txt 2>&1
SH_P3S_PAD=0 SH_P3S_NT2=0 SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 32 tests/golden/template27554.npz f32 > $O/layer_c4_old.txt 2>&1
for f in new split2 old; do echo $f; grep -h "K=576 N=64\|K=1152 N=32\|to_p3\|total" $O/layer_c4_$f.txt; done
