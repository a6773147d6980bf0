#!/bin/bash
# interleaved A/B of the two fp32 forms on one box: ms per step of bench.py (no secondary legs)
for i in 1 2 3; do
  for f in planes3 exact; do
    python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-secondary --no-roofline --f32-mma $f 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$f', round(d['ms_per_step'],4))"
  done
done
