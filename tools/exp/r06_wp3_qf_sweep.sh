#!/bin/bash
# round 6: does the plan of the three-plane weight gradient (column groups per wave) pick the fastest form?  Forced widths against the plan's own
O=gpurun_out/r06_wp3_qf; rm -rf $O; mkdir -p $O
for q in 0 4 6 8 9 11 12 16; do
  echo "== SH_WP3_QF=$q (0 = the plan's choice)" >> $O/sweep.txt
  SH_WP3_QF=$q timeout 300 python tools/wgrad_p3_probe.py 64 --reps=10 2>&1 | grep -E "^(enc|dec)" | awk '{print $1, $2, $3, $4, $(NF-1), $NF}' >> $O/sweep.txt
done
cat $O/sweep.txt
