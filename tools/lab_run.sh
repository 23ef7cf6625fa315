#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -x -k "attn or attention" 2>&1 | grep -v Warning | tail -6 | cut -c1-300
for ds in 1 0 1 0; do
  FOURM_ATTN_BWD_DS=$ds python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_breakdown_ms_per_step']; print('ds=$ds', round(d['ms_per_step'],2), 'attn_bwd', k.get('attn_bwd'), 'attn_fwd', k.get('attn_fwd'))"
done
