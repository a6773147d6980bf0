#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 600 python tools/overlap_probe.py 2>&1 | grep -v amdgpu.ids | tail -30
