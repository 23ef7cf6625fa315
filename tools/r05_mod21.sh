#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
BENCH_SHAPE_TABLE=gpurun_out/r05_shape_table_mod21_pre.txt timeout 600 python bench.py --mods mod21 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%.2f ms/step' % d['ms_per_step'], d.get('kernel_breakdown_ms_per_step'))"
head -12 gpurun_out/r05_shape_table_mod21_pre.txt
ATTN_N=256 timeout 120 python tools/attn_bench.py 2>/dev/null
