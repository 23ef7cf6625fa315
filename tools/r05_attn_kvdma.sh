#!/bin/bash
# Round 5: delta of attn_bwd128_kernel from the dO tile in LDS + whole-line loads of O - tests, then same-box A/B against -DATTN_DELTA_LDS=0.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | tail -1 > gpurun_out/r05_attn_dl_tests.txt; cat gpurun_out/r05_attn_dl_tests.txt
{ for rep in 1 2; do
    echo "== product (delta from LDS, O as whole lines)"; timeout 300 python tools/attn_bench.py
    echo "== lab build -DATTN_DELTA_LDS=0"; FOURM_HIP_LIB=$PWD/tools/bin/libfourm_hip_nodl.so timeout 300 python tools/attn_bench.py
  done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_attn_dl.txt
