#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests/test_model_gpu.py -q -m gpu --tb=short -k "graphed" 2>&1 | grep -v Warning | tail -15 | cut -c1-600
