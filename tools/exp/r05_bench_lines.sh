#!/bin/bash
# re-run of the bench lines only (bench.py changed after r05_evidence.sh; the kernel library - and with it the PMC profiles - did not)
O=gpurun_out/r05final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_f32.json 2>$O/bench_f32.err
timeout 600 python bench.py --steps 20 --warmup 5 --dtype bf16 > $O/bench_bf16.json 2>$O/bench_bf16.err
timeout 600 python bench.py --steps 20 --warmup 5 --f32-mma exact --no-secondary > $O/bench_f32_exact.json 2>$O/bench_f32_exact.err
SH_BENCH_FORCE_REDUCER=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_reducer_graph_f32.json 2>$O/bench_reducer_graph_f32.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05final/bench_*.json")):
    try:
        d=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1]); r=d.get("roofline") or {}
        print(f.split("/")[-1], round(d["ms_per_step"],4), r.get("kernel"), r.get("frac"), r.get("traffic"), r.get("traffic_gbps"))
    except Exception as e: print(f,"ERR",e)
PY
