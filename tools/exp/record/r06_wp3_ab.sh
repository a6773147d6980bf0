#!/bin/bash
# A/B of the three-plane weight gradient's work assignment switches (round 6): tools/wgrad_p3_probe.py under each setting
O=gpurun_out/r06_wp3_ab
mkdir -p $O
for cfg in "base" "SH_WP3_XSPLIT=0" "SH_WP3_DNT=1" "SH_WP3_XSPLIT=0 SH_WP3_DNT=1" $EXTRA_CFGS; do
  echo "== $cfg" >> $O/ab.txt
  if [ "$cfg" = "base" ]; then timeout 600 python tools/wgrad_p3_probe.py 64 --reps=10 2>&1 | grep -E "^(enc|dec)" >> $O/ab.txt
  else env $cfg timeout 600 python tools/wgrad_p3_probe.py 64 --reps=10 2>&1 | grep -E "^(enc|dec)" >> $O/ab.txt; fi
done
cat $O/ab.txt
