#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_headline.py -q -m gpu 2>&1 | tail -5
