#!/bin/bash
# scratch driver for one gpurun call: edit the command list, then  gpurun -- 'bash tools/lab_run.sh'
cd "$(dirname "$0")/.."
BENCH_SHAPE_TABLE=gpurun_out/mod21_shape_table.txt timeout 900 python bench.py --mods mod21 --no-traffic 2> gpurun_out/mod21.err | tail -1 > gpurun_out/mod21_bench.json
python -c "
import json; d=json.load(open('gpurun_out/mod21_bench.json')); print(d['ms_per_step'], d['value'], d['mfu'], d['kernel_breakdown_ms_per_step'])"
