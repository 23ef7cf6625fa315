#!/bin/bash
# one gpurun call: a test subset + the default step with its per-shape table.  QUICK_TESTS="tests/x.py -k expr" tools/quick.sh
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ -n "$QUICK_TESTS" ]; then timeout 900 python -m pytest $QUICK_TESTS -m gpu -x -q --tb=short 2>&1 | tail -8; fi
if [ -z "$NO_BENCH" ]; then
BENCH_SHAPE_TABLE=gpurun_out/quick_shape_table.txt timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'roofline', d['roofline']['achieved'])"
head -32 gpurun_out/quick_shape_table.txt
fi
