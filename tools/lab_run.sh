#!/bin/bash
# scratch driver for one gpurun call: edit the command list, then  gpurun -- 'bash tools/lab_run.sh'
cd "$(dirname "$0")/.."
for m in layer early layer early; do
  FOURM_DW_FLUSH=$m python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_breakdown_ms_per_step']; print('$m', round(d['ms_per_step'],2), {x: k[x] for x in ('gemm_nt/epi0','gemm_tn_multi','gemm_nt/epi2','gemm_nt/epi3')})"
done
