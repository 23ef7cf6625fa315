#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_generate_gpu.py -q -m gpu --tb=short -x 2>&1 | grep -v Warning | tail -12 | cut -c1-400
