#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu --tb=short --durations=8 2>&1 | grep -v Warning | tail -40 | cut -c1-300 > gpurun_out/lab10_tests.txt
cat gpurun_out/lab10_tests.txt
BENCH_SHAPE_TABLE=gpurun_out/shapes_mod21.txt timeout 900 python bench.py --mods mod21 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic 2>gpurun_out/bench_mod21.err | tail -1 > gpurun_out/bench_mod21.json
cut -c1-2500 gpurun_out/bench_mod21.json; tail -3 gpurun_out/bench_mod21.err; head -20 gpurun_out/shapes_mod21.txt
