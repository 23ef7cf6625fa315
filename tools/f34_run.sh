#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$(pwd)
timeout 200 python tools/f34_bench.py 2> gpurun_out/f34.err | tail -1 > gpurun_out/f34_bench.json
cat gpurun_out/f34_bench.json
rm -rf gpurun_out/prof_f34
(cd /tmp && timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_f34 -o trace -- python $ROOT/tools/f34_bench.py > /dev/null 2> $ROOT/gpurun_out/prof_f34.err)
f=$(find gpurun_out/prof_f34 -name "*kernel_stats.csv" | head -1)
head -40 "$f" | cut -c1-220 > gpurun_out/f34_kernel_stats.csv
rm -rf gpurun_out/prof_f34
grep -i "budget\|span_mask\|image_mask\|gather_emb\|latent_grad\|unpatchify\|tanh_bwd\|embed_rows" gpurun_out/f34_kernel_stats.csv | cut -c1-200
