#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_generate_gpu.py -q -m gpu --tb=short 2>&1 | grep -v Warning | grep -E "Error|assert|passed|failed|^E " | cut -c1-400 | head -30 > gpurun_out/lab9_tests.txt
cat gpurun_out/lab9_tests.txt
