#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 600 python -m pytest tests/test_optim.py tests/test_bench_tables.py -q -m gpu 2>&1 | tail -3
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 2>&1 | tail -22
timeout 300 python tools/layer_report.py 64 tests/golden/template6890.npz bf16 2>&1 | tail -8
