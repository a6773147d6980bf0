#!/bin/bash
O=gpurun_out/r05e22; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_configs.py -q -m gpu -k "latent_linear or config5" > $O/tests.txt 2>&1; tail -3 $O/tests.txt | cut -c1-250
SH_F32_MMA=planes3 timeout 300 python tools/layer_report_decode.py > $O/decode.txt 2>&1; grep "linear\|total" $O/decode.txt
SH_LIN_WIDE=0 SH_F32_MMA=planes3 timeout 300 python tools/layer_report_decode.py > $O/decode_old.txt 2>&1; grep "linear\|total" $O/decode_old.txt
SH_F32_MMA=planes3 timeout 300 python tools/bench_decode.py --latents 20480 > $O/decode.json 2>&1; tail -1 $O/decode.json | cut -c1-330
