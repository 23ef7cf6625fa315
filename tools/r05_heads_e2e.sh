#!/bin/bash
# Round 5: the per-head dense logits GEMMs (gemm_nt3 with device-side row ranges) - end-to-end tests, then same-box A/B of the step.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_b256_golden_gpu.py tests/test_trainer_gpu.py tests/test_generate_gpu.py -q -x 2>&1 | tail -2 > gpurun_out/r05_heads_e2e_tests.txt
cat gpurun_out/r05_heads_e2e_tests.txt
for v in 1 0 1 0; do
  echo "== FOURM_HEADS_DENSE=$v"
  FOURM_HEADS_DENSE=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d.get('kernel_breakdown_ms_per_step') or {}; print('%.2f ms/step' % d['ms_per_step'], 'heads nt', k.get('heads_gemm_nt (logits, dY)'))"
done 2>&1 | tee gpurun_out/r05_heads_e2e_bench.txt
echo "== mod21"
for v in 1 0; do
  FOURM_HEADS_DENSE=$v timeout 600 python bench.py --mods mod21 --steps 6 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d.get('kernel_breakdown_ms_per_step') or {}; print('FOURM_HEADS_DENSE=$v %.2f ms/step' % d['ms_per_step'], 'heads nt', k.get('heads_gemm_nt (logits, dY)'))"
done 2>&1 | tee -a gpurun_out/r05_heads_e2e_bench.txt
