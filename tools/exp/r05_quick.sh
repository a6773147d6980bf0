#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05n2; rm -rf $O; mkdir -p $O
( time timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline > $O/n2.json 2>$O/n2.err ) 2> $O/n2.time; echo "rc=$?"
tail -3 $O/n2.time; grep -c "^{" $O/n2.json; grep -h "supervisor\|Error\|error" $O/n2.err | cut -c1-200 | head -12
ps aux | grep -c "[b]ench.py"
