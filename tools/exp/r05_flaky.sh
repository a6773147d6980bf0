#!/bin/bash
cd "$(dirname "$0")/../.."
for i in 1 2 3; do timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1; done
