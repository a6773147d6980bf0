#!/bin/bash
O=gpurun_out/r05e8; rm -rf $O; mkdir -p $O
SH_F32_MMA=planes3 timeout 300 python tools/layer_report_decode.py > $O/decode_layers.txt 2>&1; cat $O/decode_layers.txt | tail -30
timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz bf16 > $O/layer_bf16.txt 2>&1; cat $O/layer_bf16.txt | tail -70
