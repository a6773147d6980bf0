#!/bin/bash
O=gpurun_out/r05e9; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_configs.py tests/test_p3.py tests/test_headline.py tests/test_editing.py -q -m gpu -x > $O/tests.txt 2>&1; tail -4 $O/tests.txt | cut -c1-300
SH_F32_MMA=planes3 timeout 300 python tools/layer_report_decode.py > $O/decode_layers.txt 2>&1; tail -14 $O/decode_layers.txt
SH_F32_MMA=planes3 timeout 300 python tools/bench_decode.py --latents 20480 > $O/decode.json 2>&1; tail -1 $O/decode.json
SH_F32_MMA=exact timeout 300 python tools/layer_report_decode.py > $O/decode_layers_exact.txt 2>&1; tail -3 $O/decode_layers_exact.txt
