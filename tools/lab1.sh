#!/bin/bash
# round-3 lab session 1: micro-benchmarks + 4-wave big-tile configurations + staggered start
mkdir -p gpurun_out; export TMPDIR=/tmp
export LD_LIBRARY_PATH=$PWD/ml-4m_amd/fourm/_lib:$LD_LIBRARY_PATH
timeout 300 tools/bin/ubench > gpurun_out/r03_ubench.txt 2>&1
timeout 200 tools/bin/gemm_lab nt 266,267,268,269,270 > gpurun_out/r03_lab_nt_4wave.txt 2>&1
timeout 300 tools/bin/gemm_lab dephase 266,267,268 > gpurun_out/r03_lab_dephase.txt 2>&1
tail -n 40 gpurun_out/r03_ubench.txt; cat gpurun_out/r03_lab_nt_4wave.txt
