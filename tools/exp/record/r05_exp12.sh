#!/bin/bash
O=gpurun_out/r05e12; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_p3.py tests/test_headline.py -q -m gpu -x > $O/tests.txt 2>&1; tail -2 $O/tests.txt | cut -c1-200
for rep in 1 2; do
  SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/layer_wpre1_$rep.txt 2>&1
  SH_KERNEL_LIB=$PWD/semantichuman_amd/lib_alt/libsh_kernels.so SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/layer_wpre0_$rep.txt 2>&1
done
for f in wpre1_1 wpre0_1 wpre1_2 wpre0_2; do echo "--- $f"; grep -h "conv_p3<\|total" $O/layer_$f.txt | grep "false, 6, false\|total"; done
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 32 tests/golden/template27554.npz f32 > $O/layer_c4_wpre1.txt 2>&1
SH_KERNEL_LIB=$PWD/semantichuman_amd/lib_alt/libsh_kernels.so SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 32 tests/golden/template27554.npz f32 > $O/layer_c4_wpre0.txt 2>&1
grep -h "conv_p3<\|total" $O/layer_c4_wpre1.txt | grep "false, 6, false\|total"; grep -h "conv_p3<\|total" $O/layer_c4_wpre0.txt | grep "false, 6, false\|total"
