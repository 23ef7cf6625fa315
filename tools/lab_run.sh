#!/bin/bash
# scratch driver for one gpurun call: edit the command list, then  gpurun -- 'bash tools/lab_run.sh'
cd "$(dirname "$0")/.."
export LD_LIBRARY_PATH=$PWD/ml-4m_amd/fourm/_lib:$LD_LIBRARY_PATH
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -x -k "gemm_tn" 2>&1 | grep -v Warning | tail -5 | cut -c1-300
for t in 256 128 256 128; do
  FOURM_TN_MULTI_TILE=$t python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_breakdown_ms_per_step']; print('tile $t', round(d['ms_per_step'],2), {x: k[x] for x in ('gemm_nt/epi0','gemm_tn_multi')})"
done
