#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
{ echo "# default library"; timeout 120 python tools/attn_bench.py 2>/dev/null
  echo "# default library, FOURM_ATTN_NONE_AS_KEYPAD=1 (the unmasked case through the key-padding instantiation)"; FOURM_ATTN_NONE_AS_KEYPAD=1 timeout 120 python tools/attn_bench.py 2>/dev/null | head -1
  for v in s1 ilp mem; do echo "# tools/bin/libfourm_hip_$v.so"; FOURM_HIP_LIB=$(pwd)/tools/bin/libfourm_hip_$v.so timeout 120 python tools/attn_bench.py 2>/dev/null; done; } > gpurun_out/r05_attn_variants.txt 2>&1
cat gpurun_out/r05_attn_variants.txt
