#!/bin/bash
# interleaved A/B of one environment knob on one box: ms per step of bench.py (planes3, no secondary legs)
# usage: r04_ab_env.sh NAME A B
for i in 1 2 3; do
  for v in $2 $3; do
    env $1=$v python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1=$v', round(d['ms_per_step'],4))"
  done
done
