#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_trainer_gpu.py tests/test_parallel_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "adamw or trainer or graph or parallel or rccl or bench_py or optimizer" --tb=short -p no:cacheprovider 2>&1 | grep -v Warning | tail -6
for i in 1 2; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); kb=d['kernel_breakdown_ms_per_step']; print('%.2f ms' % d['ms_per_step'], {k: kb[k] for k in ('adamw','grad_norm','fills','sum_of_families') if k in kb}, 'loss', d['final_loss'])"; done
