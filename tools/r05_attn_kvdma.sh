#!/bin/bash
# Round 5: K / V fragments of attn_bwd128_kernel through LDS-DMA (whole lines) - tests, then same-box A/B against -DATTN_KV_DMA=0.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | tail -1 > gpurun_out/r05_attn_kvdma_tests.txt; cat gpurun_out/r05_attn_kvdma_tests.txt
{ for rep in 1 2; do
    echo "== product (K / V by LDS-DMA)"; timeout 300 python tools/attn_bench.py
    echo "== lab build -DATTN_KV_DMA=0"; FOURM_HIP_LIB=$PWD/tools/bin/libfourm_hip_nokvdma.so timeout 300 python tools/attn_bench.py
  done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_attn_kvdma.txt
