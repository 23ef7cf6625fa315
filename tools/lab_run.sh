#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_generate_gpu.py -q -m gpu --tb=short -x -k "autoregressive" 2>&1 | grep -v Warning | tail -25 | cut -c1-400
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_subapi_gpu.py -q -m gpu --tb=short -x -k "attn or attention or subapi or forward" 2>&1 | grep -v Warning | tail -3 | cut -c1-300
