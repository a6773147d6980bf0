#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05diag; mkdir -p $O
timeout 1500 python tests/diag_long_training.py oracle 7000 2>$O/oracle.err | grep step > $O/oracle.txt
tail -3 $O/oracle.err
awk 'NR%4==1' $O/oracle.txt | cut -c1-120; tail -1 $O/oracle.txt
