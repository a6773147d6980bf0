#!/bin/bash
O=gpurun_out/r05e5; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 2400 python -m pytest tests -q -m gpu > $O/tests_all.txt 2>&1; tail -4 $O/tests_all.txt | cut -c1-300
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 64 > $O/layer_planes3.txt 2>&1
SH_F32_MMA=planes3 timeout 300 python tools/layer_report.py 32 tests/golden/template27554.npz f32 > $O/layer_c4.txt 2>&1
grep -h "wgrad_stream\|total" $O/layer_planes3.txt $O/layer_c4.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_f32.json 2>$O/bench_f32.err; tail -3 $O/bench_f32.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r05e5/bench_f32.json").read().splitlines() if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"]); print(json.dumps(d["roofline"])[:600]); print(json.dumps(d.get("roofline_matrix_family"))[:400]); print(json.dumps(d["whole_step"])[:900])
for k,v in d["secondary"].items(): print(k, v.get("ms_per_step"), v.get("p50_batch_ms"), v.get("mean_batch_ms"), v.get("error"))
PY
