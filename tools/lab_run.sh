#!/bin/bash
# scratch driver for one gpurun call: edit the command list, then  gpurun -- 'bash tools/lab_run.sh'
cd "$(dirname "$0")/.."
export LD_LIBRARY_PATH=$PWD/ml-4m_amd/fourm/_lib:$LD_LIBRARY_PATH TMPDIR=/tmp
mkdir -p gpurun_out
# fused epilogues (residual add, SwiGLU backward) on the kernels with TWO workgroups per CU (gemm.hip configurations 2 / 8 / 11: 128 x 256 tiles, K-step 32, 72 KB of LDS)
# against one workgroup per CU (10, 12) and the lock-step kernels (2001 = nt4 / nt3 by shape)
timeout 400 tools/bin/gemm_lab nt 2001,2,8,11,10,12 2>&1 | tee gpurun_out/lab_fused_2wg.txt
