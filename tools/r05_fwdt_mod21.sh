#!/bin/bash
# Round 5: attention tests on the current tree, then 4M-L mod21 with and without the whole-tile forward (same box).
cd "$(dirname "$0")/.." && mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | tail -1
for t in 1 0 1 0; do
  echo "== FOURM_ATTN_FWD_T=$t"
  FOURM_ATTN_FWD_T=$t BENCH_SHAPE_TABLE=gpurun_out/r05_fwdt_mod21_table_$t.txt timeout 600 python bench.py --mods mod21 --steps 6 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d.get('kernel_breakdown_ms_per_step') or {}; print('%.2f ms/step' % d['ms_per_step'], 'attn_fwd', k.get('attn_fwd'), 'attn_bwd', k.get('attn_bwd'))"
done 2>&1 | tee gpurun_out/r05_fwdt_mod21.txt
