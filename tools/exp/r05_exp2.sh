#!/bin/bash
O=gpurun_out/r05e2; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_p3.py tests/test_configs.py tests/test_wgrad_thin.py -q -m gpu > $O/test_parity.txt 2>&1; tail -4 $O/test_parity.txt
timeout 900 python -m pytest tests/test_headline.py -q -m gpu -s > $O/test_headline.txt 2>&1; tail -5 $O/test_headline.txt
for rep in 1 2; do for x in 0 1; do
  SH_WS_XCD_RANGES=$x SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/layer_planes3_xg${x}_$rep.txt 2>/dev/null
  grep -h "wgrad_stream" $O/layer_planes3_xg${x}_$rep.txt | awk -v x=$x '{s+=$(NF-1)} END {print "xg" x " wgrad_stream sum", s}'
  grep -h total $O/layer_planes3_xg${x}_$rep.txt
done; done
timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary > $O/bench_xg1.json 2>$O/bench_xg1.err
SH_WS_XCD_RANGES=0 timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary > $O/bench_xg0.json 2>$O/bench_xg0.err
python - <<'PY'
import json
for f in ("xg1","xg0"):
    try:
        d=json.loads([l for l in open("gpurun_out/r05e2/bench_%s.json"%f).read().splitlines() if l.startswith("{")][-1]); print(f, d["ms_per_step"], d["value"])
    except Exception as e: print(f, "ERR", e)
PY
