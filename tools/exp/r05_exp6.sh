#!/bin/bash
O=gpurun_out/r05e6; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_p3.py -q -m gpu -s -k "latent_fc" > $O/tests_fc.txt 2>&1; grep -h "^FC\|passed\|failed\|Error" $O/tests_fc.txt | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "latent_linear or conv_kernels or sequencing or golden or full_size" > $O/tests2.txt 2>&1; tail -3 $O/tests2.txt | cut -c1-300
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/layer_planes3.txt 2>&1
SH_LIN_X3=0 SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/layer_planes3_nox3.txt 2>&1
SH_P3_WG_SPLIT3=1 SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/layer_planes3_wgs3.txt 2>&1
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 32 tests/golden/template27554.npz f32 > $O/layer_c4.txt 2>&1
echo "--- x3"; grep -h "linear\|split_reduce\|wgrad_stream\|total" $O/layer_planes3.txt
echo "--- no x3"; grep -h "linear\|total" $O/layer_planes3_nox3.txt
echo "--- wg split3"; grep -h "wgrad\|total\|spmm" $O/layer_planes3_wgs3.txt
echo "--- c4"; grep -h "wgrad_stream\|linear\|total" $O/layer_c4.txt
