#!/bin/bash
# Round 5: O / dQ / dK / dV of the whole-tile attention kernels as whole-line stores (staged through LDS) - tests on the product library, then
# same-box A/B against the lab build without them (tools/r05_attn_variants.sh nostage "-DATTN_STAGED_STORES=0").
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | tail -1 > gpurun_out/r05_attn_staged_tests.txt; cat gpurun_out/r05_attn_staged_tests.txt
{ for rep in 1 2; do
    echo "== product (staged whole-line stores)"; timeout 300 python tools/attn_bench.py
    echo "== lab build -DATTN_STAGED_STORES=0"; FOURM_HIP_LIB=$PWD/tools/bin/libfourm_hip_nostage.so timeout 300 python tools/attn_bench.py
  done
  echo "== product, FOURM_ATTN_BWD_OV=0 (attn_bwd128_kernel for every mask)"; FOURM_ATTN_BWD_OV=0 timeout 300 python tools/attn_bench.py
  echo "== product, FOURM_ATTN_BWD_OV=1 (attn_bwd128o_kernel for every mask)"; FOURM_ATTN_BWD_OV=1 timeout 300 python tools/attn_bench.py
  echo "== N = 256: product"; ATTN_N=256 timeout 300 python tools/attn_bench.py
  echo "== N = 256: lab build -DATTN_STAGED_STORES=0"; ATTN_N=256 FOURM_HIP_LIB=$PWD/tools/bin/libfourm_hip_nostage.so timeout 300 python tools/attn_bench.py
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_attn_staged.txt
