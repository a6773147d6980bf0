#!/bin/bash
# what the driver does at round end, on one box: the GPU suite, smoke(), the default bench line
cd "$(dirname "$0")/../.."
O=gpurun_out/r05drv; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
( time timeout 1200 python bench.py > $O/bench.json 2>$O/bench.err ) 2> $O/bench.time; echo "bench rc=$?"; cat $O/bench.time | tail -3
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r05drv/bench.json") if l.startswith("{")][-1])
print(d["metric"], d["value"], d["unit"], d["ms_per_step"], d["steps"], d["warmup"], d["dtype"], d["vs_baseline"], d["roofline"]["kernel"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
PY
