#!/bin/bash
# round 6: is the ragged bf16 backward-data at least as close to the fp32 gradients as the dense form?  (one step, batch 16 and 64)
O=gpurun_out/r06bfe; rm -rf $O; mkdir -p $O
for B in 16 64; do for cfg in 1 0; do
  SH_BF16_RAGGED=$cfg timeout 300 python tools/exp/bf16_grad_err.py $B > $O/err_b${B}_rag$cfg.txt 2>&1; echo "B=$B ragged=$cfg"; grep -E "conv.*weight|SUM" $O/err_b${B}_rag$cfg.txt
done; done
for cfg in 1 0; do SH_BF16_RAGGED=$cfg timeout 600 python -m pytest tests/test_bf16.py -m gpu -q -k "matched_l2" 2>&1 | grep -E "AssertionError: |passed|failed" | head -3; done
