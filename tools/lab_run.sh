#!/bin/bash
# scratch driver for one gpurun call: edit the command list, then  gpurun -- 'bash tools/lab_run.sh'
cd "$(dirname "$0")/.."
python tools/shadow_bench.py 2>&1 | grep -v Warn | tail -3
