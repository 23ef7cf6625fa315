#!/bin/bash
# Round 5, attention backward: kernel tests, stand-alone timing of the general and the 128 x 128 kernel, counters, whole-step A/B.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -k "attention or micro or b_mod7 or ti_mod7" --tb=short -p no:cacheprovider > gpurun_out/r05_attn_pytest.txt 2>&1
tail -15 gpurun_out/r05_attn_pytest.txt
{ for v in 0 1; do echo "# FOURM_ATTN_BWD_V2=$v"; FOURM_ATTN_BWD_V2=$v timeout 120 python tools/attn_bench.py 2>/dev/null; done; } > gpurun_out/r05_attn_bench.txt 2>&1
cat gpurun_out/r05_attn_bench.txt
timeout 900 bash tools/ab_env.sh "FOURM_ATTN_BWD_V2=0" "FOURM_ATTN_BWD_V2=1" > gpurun_out/r05_ab_attn_bwd_v2.txt 2>&1
cat gpurun_out/r05_ab_attn_bwd_v2.txt
timeout 600 bash tools/pmc_attn_run.sh > gpurun_out/r05_pmc_attn.txt 2>&1
tail -3 gpurun_out/r05_pmc_attn.txt
