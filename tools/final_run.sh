#!/bin/bash
# Round-end measurement set on one MI355X: full GPU suite (parity log), the default bench line, rocprofv3 kernel stats of the bench
# command, the VQ bench line.  Everything lands under gpurun_out/ (copy what is to be judged into profiles/).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/parity.jsonl
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1500 python -m pytest tests -v -m gpu -x --tb=short > gpurun_out/final_pytest_full.txt 2>&1     # (-v into a file: a cut-off run still shows how far it got)
grep -v Warning gpurun_out/final_pytest_full.txt | tail -12 | cut -c1-300 > gpurun_out/final_pytest.txt
tail -3 gpurun_out/final_pytest.txt
BENCH_SHAPE_TABLE=gpurun_out/final_shape_table.txt timeout 900 python bench.py 2> gpurun_out/final_bench.err | tail -1 > gpurun_out/final_bench.json
cut -c1-600 gpurun_out/final_bench.json
rm -rf gpurun_out/prof_r02
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_r02 -o trace -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-traffic > $ROOT/gpurun_out/final_bench_under_rocprof.json 2> $ROOT/gpurun_out/prof_r02.err)
f=$(find gpurun_out/prof_r02 -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/final_kernel_stats.csv
rm -rf gpurun_out/prof_r02
head -12 gpurun_out/final_kernel_stats.csv | cut -c1-160
timeout 600 python bench.py --workload vq 2> gpurun_out/final_bench_vq.err | tail -1 > gpurun_out/final_bench_vq.json
cut -c1-400 gpurun_out/final_bench_vq.json
