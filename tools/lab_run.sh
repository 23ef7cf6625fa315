#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out; rm -rf gpurun_out/prof_vq
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_vq -o trace -- python $ROOT/bench.py --workload vq --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-traffic > $ROOT/gpurun_out/vq_under_rocprof.json 2> $ROOT/gpurun_out/prof_vq.err)
f=$(find gpurun_out/prof_vq -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/vq_kernel_stats.csv; rm -rf gpurun_out/prof_vq
head -8 gpurun_out/vq_kernel_stats.csv | cut -c1-150
BENCH_SHAPE_TABLE=gpurun_out/mod21_shape_table.txt timeout 900 python bench.py --mods mod21 --no-traffic 2> gpurun_out/mod21.err | tail -1 > gpurun_out/mod21_bench.json
cut -c1-300 gpurun_out/mod21_bench.json
