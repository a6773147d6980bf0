#!/bin/bash
set -u
O=gpurun_out/r03e22; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_sem -o sem --output-format csv -- python3 tools/bench_semantic.py --graph --steps 50 > $O/sem.json 2>$O/sem.err
cp $(find $O/prof_sem -name "*kernel_stats.csv" | head -1) $O/sem_kernel_stats.csv; rm -rf $O/prof_sem
cat $O/sem.json
