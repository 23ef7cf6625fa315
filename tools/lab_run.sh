#!/bin/bash
# scratch driver for one gpurun call: edit the command list, then  gpurun -- 'bash tools/lab_run.sh'
cd "$(dirname "$0")/.."
export LD_LIBRARY_PATH=$PWD/ml-4m_amd/fourm/_lib:$LD_LIBRARY_PATH TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 tools/bin/gemm_lab nt ${CFGS:-1003,2001,2003,2005,2041} > gpurun_out/lab_nt4.txt 2>&1
cat gpurun_out/lab_nt4.txt
