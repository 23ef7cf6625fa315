#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "heads_dense or gemm_grouped or gemm_nt" 2>&1 | tail -3
{ timeout 300 python tools/heads_bench.py 2>&1 | grep -v amdgpu.ids | head -6
  echo "== mod21-like: 21 heads"; VOCABS=16384,8192,8192,4096,8192,30000,30000,1024,1024,8192,8192,8192,512,512,30000,8192,8192,16384,8192,4096,1024 timeout 300 python tools/heads_bench.py 2>&1 | grep -v amdgpu.ids | head -6; } | tee gpurun_out/r05_heads_lab.txt
