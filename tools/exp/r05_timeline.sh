#!/bin/bash
# one replayed step as a timeline: kernel start / end from rocprofv3's kernel trace, gaps between consecutive kernels
O=gpurun_out/r05tl; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 600 rocprofv3 --kernel-trace -d $O/tr -o t --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline "$@" > $O/bench.json 2>$O/bench.err
T=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $T > $O/timeline.txt
rm -rf $O/tr
tail -30 $O/timeline.txt
