#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu --tb=long -x -k graphed 2>&1 | grep -v Warning | tail -60 | cut -c1-300
