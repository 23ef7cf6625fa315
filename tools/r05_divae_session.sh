#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_divae.py tests/test_vq_euclid.py tests/test_masking.py -m gpu -q -s --tb=short -p no:cacheprovider > gpurun_out/r05_divae_pytest.txt 2>&1
grep -v Warning gpurun_out/r05_divae_pytest.txt | tail -40
