#!/bin/bash
# one GPU-box call: GEMM lab (ablations)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for abl in 0 5 7 21 23 13; do echo "== ablate $abl"; FOURM_NT_ABLATE=$abl tools/bin/gemm_lab nt 266,268; done
} > gpurun_out/lab2.txt 2>&1
cat gpurun_out/lab2.txt
