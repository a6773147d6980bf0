#!/bin/bash
O=gpurun_out/r05e3; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 2400 python -m pytest tests -q -m gpu -rs > $O/tests_all.txt 2>&1; tail -40 $O/tests_all.txt | cut -c1-300
