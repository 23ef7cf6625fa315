#!/bin/bash
# scratch driver for one gpurun call: edit the command list, then  gpurun -- 'bash tools/lab_run.sh'
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_parallel_gpu.py -q -m gpu --tb=short -x -k "codebook" 2>&1 | grep -v Warning | tail -12 | cut -c1-400
