#!/bin/bash
# does the three-plane weight gradient wait for bytes from beyond L2?  Real tables against a perfectly local one, with HBM fetch sizes
O=gpurun_out/r06_wp3_loc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for v in real local; do
  F=""; [ $v = local ] && F="--local-table"
  timeout 600 python tools/wgrad_p3_probe.py 64 --reps=10 $F 2>&1 | grep -E "^(enc|dec)" > $O/time_$v.txt
  timeout 600 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/p_$v -o p --output-format csv -- python3 tools/wgrad_p3_probe.py 64 --reps=2 $F > $O/p_$v.log 2>&1
  python3 tools/pmc_table.py $O/table_$v.txt $O/p_$v > /dev/null 2>&1
  rm -rf $O/p_$v
  echo "== $v"; cat $O/time_$v.txt; grep -A1 "^wgrad_p3" $O/table_$v.txt
done
