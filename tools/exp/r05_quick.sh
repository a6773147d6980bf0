#!/bin/bash
cd "$(dirname "$0")/../.."
for v in 1 0 1 0; do
  echo "== SH_SLAB_REDUCE_VEC=$v"
  SH_SLAB_REDUCE_VEC=$v SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 2>/dev/null | grep -E "slab_reduce|total"
done
SH_SLAB_REDUCE_VEC=1 timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz bf16 2>/dev/null | grep -E "slab_reduce|total"
SH_SLAB_REDUCE_VEC=0 timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz bf16 2>/dev/null | grep -E "slab_reduce|total"
SH_SLAB_REDUCE_VEC=1 SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 32 tests/golden/template27554.npz f32 2>/dev/null | grep -E "slab_reduce|total"
SH_SLAB_REDUCE_VEC=0 SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 32 tests/golden/template27554.npz f32 2>/dev/null | grep -E "slab_reduce|total"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_headline.py tests/test_bf16.py -q -m gpu -x 2>&1 | tail -3
