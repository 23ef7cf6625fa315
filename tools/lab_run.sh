#!/bin/bash
# scratch driver for one gpurun call: edit the command list, then  gpurun -- 'bash tools/lab_run.sh'
cd "$(dirname "$0")/.."
export LD_LIBRARY_PATH=$PWD/ml-4m_amd/fourm/_lib:$LD_LIBRARY_PATH TMPDIR=/tmp
mkdir -p gpurun_out
echo "old"; timeout 200 tools/bin/gemm_lab tnmulti 32768 1 1 0 0 | cut -c60-140
for c in 4 8 16 32 64 128; do for cb in 4 16 48; do echo "t4 c=$c cb=$cb"; timeout 200 tools/bin/gemm_lab tnmulti 32768 1 1 1 0 $c $cb | cut -c60-140; done; done 2>&1 | tee gpurun_out/lab_tn4_c.txt
