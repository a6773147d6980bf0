#!/bin/bash
O=gpurun_out/r05e10; rm -rf $O; mkdir -p $O
for b in 64 128 256 512 1024; do
  SH_F32_MMA=planes3 timeout 300 python tools/layer_report_decode.py $b > $O/decode_layers_$b.txt 2>&1
  echo "--- B=$b"; grep -h "linear\|spmm\|conv\|total" $O/decode_layers_$b.txt | awk -v b=$b '{printf "%-40s %10.1f  per-64: %8.1f\n", $1 " " $2 " " $3, $(NF-1)+0 > 0 ? $(NF-1) : $NF, ($(NF-1)+0 > 0 ? $(NF-1) : $NF) * 64 / b}'
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "c_abi or latent_linear" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
