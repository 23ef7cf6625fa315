#!/bin/bash
# rocprofv3 kernel stats of the bench command (3 timed + 1 warm-up step = 4 steps, like profiles/r01_*): summary -> gpurun_out/prof_r02
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -rf gpurun_out/prof_r02
export TMPDIR=/tmp
ROOT=$(pwd)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_r02 -o trace -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-profile > $ROOT/gpurun_out/prof_r02_bench.json 2> $ROOT/gpurun_out/prof_r02.err)
f=$(find gpurun_out/prof_r02 -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/prof_r02_kernel_stats.csv
find gpurun_out/prof_r02 -name "*kernel_trace.csv" -delete
find gpurun_out/prof_r02 -name "*.db" -delete
head -40 gpurun_out/prof_r02_kernel_stats.csv | cut -c1-200
