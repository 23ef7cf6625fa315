#!/bin/bash
# same-box A/B of the whole train step: FOURM_NT3=0 (previous NT kernels) vs the default, interleaved
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
  FOURM_NT3=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 > gpurun_out/ab_old_$rep.json
  BENCH_SHAPE_TABLE=gpurun_out/ab_shapes_$rep.txt python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 > gpurun_out/ab_new_$rep.json
done
python - <<'PY'
import json
for rep in (1, 2):
    a = json.load(open(f"gpurun_out/ab_old_{rep}.json")); b = json.load(open(f"gpurun_out/ab_new_{rep}.json"))
    print(f"rep {rep}: old {a['ms_per_step']:.2f} ms   new {b['ms_per_step']:.2f} ms")
    ka, kb = a.get("kernel_breakdown_ms_per_step", {}), b.get("kernel_breakdown_ms_per_step", {})
    print("   " + "  ".join(f"{k}: {ka.get(k, 0):.2f}->{kb.get(k, 0):.2f}" for k in kb))
PY
cat gpurun_out/ab_shapes_2.txt
