#!/bin/bash
# same-box A/B of the whole train step: round-1 tree (tools/bin/r01, built from commit 5db2a4b) vs the working tree, interleaved
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do
  (cd tools/bin/r01 && python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 > ../../../gpurun_out/ab_r01_$rep.json)
  BENCH_SHAPE_TABLE=gpurun_out/ab_shapes_$rep.txt python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 > gpurun_out/ab_cur_$rep.json
done
python - <<'PY'
import json
for rep in (1, 2):
    a = json.load(open(f"gpurun_out/ab_r01_{rep}.json")); b = json.load(open(f"gpurun_out/ab_cur_{rep}.json"))
    print(f"rep {rep}: r01 {a['ms_per_step']:.2f} ms   current {b['ms_per_step']:.2f} ms")
    ka, kb = a["kernel_breakdown_ms_per_step"], b["kernel_breakdown_ms_per_step"]
    print("   " + "  ".join(f"{k}: {ka.get(k, 0):.2f}->{kb.get(k, 0):.2f}" for k in kb))
PY
