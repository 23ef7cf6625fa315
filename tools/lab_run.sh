#!/bin/bash
# scratch driver for one gpurun call: edit the command list, then  gpurun -- 'bash tools/lab_run.sh'
cd "$(dirname "$0")/.."
export LD_LIBRARY_PATH=$PWD/ml-4m_amd/fourm/_lib:$LD_LIBRARY_PATH
timeout 300 tools/bin/gemm_lab tnmulti 2>&1 | tail -2
timeout 300 tools/bin/gemm_lab cold 2>&1 | tail -3
python tools/attn_bench.py 2>&1 | grep -v Warn | tail -3
