#!/bin/bash
# Round 5: the whole-tile attention forwards (attn_fwd128_kernel, attn_fwdt_kernel) - tests, then the same-box A/B against the general kernel.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | tail -5 > gpurun_out/r05_attn_fwd128_tests.txt
cat gpurun_out/r05_attn_fwd128_tests.txt
FOURM_ATTN_FWD_DB=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention_fwd128" 2>&1 | tail -1
{
  echo "== N = 128: default (attn_fwd128_kernel)"; timeout 300 python tools/attn_bench.py
  echo "== N = 128: FOURM_ATTN_FWD_V2=0 FOURM_ATTN_FWD_T=0 (attn_fwd_kernel)"; FOURM_ATTN_FWD_V2=0 FOURM_ATTN_FWD_T=0 timeout 300 python tools/attn_bench.py
  echo "== N = 128: FOURM_ATTN_FWD_T=2 (attn_fwdt_kernel with one tile)"; FOURM_ATTN_FWD_T=2 timeout 300 python tools/attn_bench.py
  echo "== N = 256: default (attn_fwdt_kernel, one K/V buffer, 3 workgroups per CU)"; ATTN_N=256 timeout 300 python tools/attn_bench.py
  echo "== N = 256: FOURM_ATTN_FWD_DB=1 (attn_fwdt_kernel, two K/V buffers, 2 workgroups per CU)"; ATTN_N=256 FOURM_ATTN_FWD_DB=1 timeout 300 python tools/attn_bench.py
  echo "== N = 256: FOURM_ATTN_FWD_T=0 (attn_fwd_kernel)"; ATTN_N=256 FOURM_ATTN_FWD_T=0 timeout 300 python tools/attn_bench.py
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_attn_fwd128.txt
