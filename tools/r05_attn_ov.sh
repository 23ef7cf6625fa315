#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
FOURM_ATTN_BWD_OV=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "attention and 128-128" --tb=short -p no:cacheprovider 2>&1 | grep -v Warning | tail -8
{ for v in 0 1; do echo "# FOURM_ATTN_BWD_OV=$v"; FOURM_ATTN_BWD_OV=$v timeout 120 python tools/attn_bench.py 2>/dev/null; done; } > gpurun_out/r05_attn_ov.txt 2>&1
cat gpurun_out/r05_attn_ov.txt
timeout 900 bash tools/ab_env.sh "FOURM_ATTN_BWD_OV=0" "FOURM_ATTN_BWD_OV=1" > gpurun_out/r05_ab_attn_ov.txt 2>&1
cat gpurun_out/r05_ab_attn_ov.txt
