#!/bin/bash
# rocprofv3 kernel stats of the headline bench command only (no tests): gpurun_out/step_kernel_stats.csv
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$(pwd)
rm -rf gpurun_out/prof_step
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_step -o trace -- env $1 python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-traffic --no-extras > $ROOT/gpurun_out/step_under_rocprof.json 2> $ROOT/gpurun_out/prof_step.err)
f=$(find gpurun_out/prof_step -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/step_kernel_stats.csv
t=$(find gpurun_out/prof_step -name "*kernel_trace.csv" | head -1); gzip -c "$t" > gpurun_out/step_kernel_trace.csv.gz
rm -rf gpurun_out/prof_step
cut -c1-300 gpurun_out/step_under_rocprof.json
head -5 gpurun_out/step_kernel_stats.csv | cut -c1-200
