#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
for c in 0 3 4; do echo "== FOURM_HEADS_NT_CFG=$c"; FOURM_HEADS_NT_CFG=$c timeout 300 python tools/heads_bench.py 2>&1 | grep -v amdgpu.ids | head -3; done | tee gpurun_out/r05_heads_lab2.txt
