#!/bin/bash
# Round 5: the 128 x 128 attention forward (attn_fwd128_kernel) - tests, then the same-box A/B against the general kernel.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | tail -5 > gpurun_out/r05_attn_fwd128_tests.txt
cat gpurun_out/r05_attn_fwd128_tests.txt
{
  for rep in 1 2; do
    echo "== FOURM_ATTN_FWD_V2=1 (attn_fwd128_kernel)"; FOURM_ATTN_FWD_V2=1 timeout 300 python tools/attn_bench.py
    echo "== FOURM_ATTN_FWD_V2=0 (attn_fwd_kernel)";    FOURM_ATTN_FWD_V2=0 timeout 300 python tools/attn_bench.py
  done
} 2>&1 | tee gpurun_out/r05_attn_fwd128.txt
