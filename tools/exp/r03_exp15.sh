#!/bin/bash
# Adam riders: parity + A/B
set -u
O=gpurun_out/r03e15; mkdir -p $O
timeout 600 python -m pytest tests/test_optim.py -x -q -m gpu > $O/tests_optim.txt 2>&1
tail -3 $O/tests_optim.txt
for r in on off; do
  timeout 300 python bench.py --steps 200 --warmup 20 --no-secondary --adam-rider $r > $O/bench_rider_$r.json 2> $O/bench_rider_$r.err
done
for g in 1500 2500 4000 6000; do
  SH_RIDER_GBPS=$g timeout 300 python bench.py --steps 200 --warmup 20 --no-secondary --no-roofline > $O/bench_gbps_$g.json 2> $O/bench_gbps_$g.err
done
for b in 128 256 1024; do
  SH_RIDER_BLOCKS=$b timeout 300 python bench.py --steps 200 --warmup 20 --no-secondary --no-roofline > $O/bench_blocks_$b.json 2> $O/bench_blocks_$b.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03e15/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], j["ms_per_step"], j.get("launch"))
    except Exception as e: print(f, "ERR", e)
PY
timeout 900 python -m pytest tests -x -q -m gpu > $O/tests_all.txt 2>&1; tail -3 $O/tests_all.txt
