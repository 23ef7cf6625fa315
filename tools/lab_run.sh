#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemm_tn or grouped" --tb=short 2>&1 | grep -v Warning | tail -5 > gpurun_out/lab7_tests.txt
cat gpurun_out/lab7_tests.txt
bash tools/ab_bench.sh
