#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_vq.py tests/test_vqvae.py tests/test_divae.py -m gpu -q -s --tb=short -p no:cacheprovider 2>&1 | grep -v Warning | tail -25
{ for v in 0 1; do echo "# FOURM_VQ_SPLIT3=$v"; FOURM_VQ_SPLIT3=$v timeout 300 python bench.py --workload vq --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%.0f images/s  %.2f ms/batch' % (d['value'], d['ms_per_step']), d.get('kernel_breakdown_ms_per_step'))"; done; } > gpurun_out/r05_ab_vq_split3.txt 2>&1
cat gpurun_out/r05_ab_vq_split3.txt
timeout 600 python tools/divae_bench.py 8 25 > gpurun_out/r05_divae_bench.txt 2>&1; tail -3 gpurun_out/r05_divae_bench.txt
