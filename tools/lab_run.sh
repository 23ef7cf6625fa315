#!/bin/bash
# scratch driver for one gpurun call: edit the command list, then  gpurun -- 'bash tools/lab_run.sh'
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -x -k "gemm_tn" 2>&1 | grep -v Warning | tail -3 | cut -c1-300
for t in 1 0 1 0; do
  FOURM_TN_BANDS=$t python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_breakdown_ms_per_step']; print('bands $t', round(d['ms_per_step'],2), {x: k[x] for x in ('gemm_nt/epi0','gemm_tn_multi')})"
done
