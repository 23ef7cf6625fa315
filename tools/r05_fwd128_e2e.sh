#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_generate_gpu.py -q -x 2>&1 | tail -4 > gpurun_out/r05_fwd128_e2e_tests.txt
cat gpurun_out/r05_fwd128_e2e_tests.txt
for v in 1 0 1 0; do
  echo "== FOURM_ATTN_FWD_V2=$v"
  FOURM_ATTN_FWD_V2=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-extras --no-kernel-profile 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%.2f ms/step' % d['ms_per_step'])"
done 2>&1 | tee gpurun_out/r05_fwd128_e2e_bench.txt
