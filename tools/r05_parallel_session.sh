#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_parallel_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | grep -v Warning | tail -30
