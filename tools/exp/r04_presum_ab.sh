#!/bin/bash
# planes3: pre-summed rows imaged by their producers (SH_P3_PRESUM_IMG=1) vs split by the backward-data kernel (default)
cd "$(dirname "$0")/../.."
python tools/p3_probe.py 64 --bwd --reps=10 2>&1 | grep -E "bwd"
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
for v in 0 1 0 1; do
  SH_P3_PRESUM_IMG=$v python bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('presum_img=$v', r['ms_per_step'], r['value'], r.get('hip_kernel_ms_per_step'), r['roofline']['kernel'], r['roofline']['frac'])"
done
SH_F32_MMA=planes3 python tools/layer_report.py 64 2>/dev/null | grep -E "wgrad_stream|spmm|conv_p3|total"
