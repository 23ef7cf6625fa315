#!/bin/bash
# one GPU-box call: flat-kernel tests + GEMM lab A/B (flat vs tile-at-a-time)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "flat or nt_plain or tile_configs or attention" --tb=short 2>&1 | grep -v Warning | tail -40 > gpurun_out/lab3_tests.txt
{
echo "== flat (265) vs tiled (265 + bit 28 = 268435721)"; tools/bin/gemm_lab nt 265,268435721
} > gpurun_out/lab3.txt 2>&1
cat gpurun_out/lab3_tests.txt gpurun_out/lab3.txt
