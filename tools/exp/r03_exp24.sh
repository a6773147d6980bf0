#!/bin/bash
timeout 600 python tools/exp/sem_ops_probe.py > gpurun_out/r03e24_probe.txt 2>&1; tail -80 gpurun_out/r03e24_probe.txt
