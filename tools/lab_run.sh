#!/bin/bash
# scratch driver for one gpurun call: edit the command list, then  gpurun -- 'bash tools/lab_run.sh'
cd "$(dirname "$0")/.."
export LD_LIBRARY_PATH=$PWD/ml-4m_amd/fourm/_lib:$LD_LIBRARY_PATH TMPDIR=/tmp
mkdir -p gpurun_out
timeout 500 tools/bin/gemm_lab ldpad 2001 0,64,128 2>&1 | tee gpurun_out/lab_ldpad.txt
