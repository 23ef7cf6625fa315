#!/bin/bash
# One GPU-box session: hardware fact probe + kernel numerics.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -5 > gpurun_out/box.txt
nproc >> gpurun_out/box.txt; lscpu | grep "Model name" >> gpurun_out/box.txt
timeout 120 tools/bin/probe_gfx950 > gpurun_out/probe.txt 2>&1
grep -E "AS_ASSUMED|LANE_LINEAR" gpurun_out/probe.txt
timeout ${PYTEST_TIMEOUT:-1500} python -m pytest ${PYTEST_ARGS:-tests/test_kernels_gpu.py} -m gpu -q -n 2 --timeout 300 --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -5
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu.log | head -60
