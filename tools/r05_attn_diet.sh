#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_fullsize_l_gpu.py -m gpu -q -x -k "attention or l_mod21 or l_like or micro or zero_attn or fullsize" --tb=short -p no:cacheprovider 2>&1 | grep -v Warning | tail -8
ATTN_N=256 timeout 120 python tools/attn_bench.py 2>/dev/null
BENCH_SHAPE_TABLE=gpurun_out/r05_shape_table_mod21.txt timeout 600 python bench.py --mods mod21 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%.2f ms/step' % d['ms_per_step'], {k: v for k, v in d.get('kernel_breakdown_ms_per_step').items() if 'attn' in k})"
