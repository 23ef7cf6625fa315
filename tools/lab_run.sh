#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_model_gpu.py -q -m gpu -k "fp32_verification" --tb=short -x 2>&1 | grep -v Warning | tail -40 > gpurun_out/lab8_tests.txt
cat gpurun_out/lab8_tests.txt; grep fp32_mode gpurun_out/parity.jsonl | tail -6
