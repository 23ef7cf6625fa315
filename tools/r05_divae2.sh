#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_divae.py -m gpu -q -s --tb=short -p no:cacheprovider 2>&1 | grep -v Warning | tail -12
timeout 600 python tools/divae_bench.py 8 25 > gpurun_out/r05_divae_bench2.txt 2>&1; tail -3 gpurun_out/r05_divae_bench2.txt
timeout 600 python tools/divae_bench.py 32 10 > gpurun_out/r05_divae_bench2_b32.txt 2>&1; tail -3 gpurun_out/r05_divae_bench2_b32.txt
