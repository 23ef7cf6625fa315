#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_parallel_gpu.py -q -m gpu --tb=short 2>&1 | grep -v Warning | tail -12 | cut -c1-600
