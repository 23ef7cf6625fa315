#!/bin/bash
# Round-end measurement set on one MI355X: smoke, full GPU suite (parity log), the default bench line (headline + extra.vq + extra.mod21 + extra.divae),
# rocprofv3 kernel stats of the bench command, lab tables of the GEMM kernels.  Everything lands under gpurun_out/ (copy what is to be
# judged into profiles/).
cd "$(dirname "$0")/.."
TAG=${TAG:-r06_final}
mkdir -p gpurun_out; rm -f gpurun_out/parity.jsonl
export TMPDIR=/tmp
ROOT=$(pwd)
export LD_LIBRARY_PATH=$ROOT/ml-4m_amd/fourm/_lib:$LD_LIBRARY_PATH
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
if [ -z "$SKIP_TESTS" ]; then
  timeout 1500 python -m pytest tests -v -m gpu -x --tb=short --durations=15 > gpurun_out/${TAG}_pytest_full.txt 2>&1     # (-v into a file: a cut-off run still shows how far it got)
  grep -v Warning gpurun_out/${TAG}_pytest_full.txt | tail -30 | cut -c1-300 > gpurun_out/${TAG}_pytest.txt
  tail -3 gpurun_out/${TAG}_pytest.txt
  cp gpurun_out/parity.jsonl profiles/r06_parity.jsonl      # bench.py's `parity` record reads the newest profiles/rNN_parity.jsonl
fi
BENCH_SHAPE_TABLE=gpurun_out/${TAG}_shape_table.txt timeout 900 python bench.py 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json
cut -c1-400 gpurun_out/${TAG}_bench.json
rm -rf gpurun_out/prof_final
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_final -o trace -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-traffic --no-extras > $ROOT/gpurun_out/${TAG}_bench_under_rocprof.json 2> $ROOT/gpurun_out/prof_final.err)
f=$(find gpurun_out/prof_final -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/prof_final
head -12 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160
# the lab binary is rebuilt against the current header (fm_gemm_nt_args grows with the ABI; a stale binary would pass a short struct)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -I include tools/gemm_lab.cpp -L ml-4m_amd/fourm/_lib -lfourm_hip -o tools/bin/gemm_lab 2>/dev/null
{ echo "# tools/gemm_lab nt 1003,2001,2041,2081: gemm_nt3 by shape (round 5), gemm_nt4 where it applies (the default), the same without its epilogue / with its stores dropped by the bounds check";
  timeout 200 tools/bin/gemm_lab nt 1003,2001,2041,2081 | sed "s/\[c[0-9]*: [0-9]* of [0-9]* halfwords differ from c[0-9]*\]//g";
  echo "# round-2 automatic choice (265), round-2 256x256 ping-pong (268), gemm_nt3 256-wide / 192-wide (1001 / 1002)";
  timeout 200 tools/bin/gemm_lab nt 265,268,1001,1002;
  echo "# T(K) at fixed M, N (operands with leading dimension 6144)";
  timeout 200 tools/bin/gemm_lab ksweep 266,1001,1002;
  echo "# all dW GEMMs of a layer: one fm_gemm_tn launch each vs ONE fm_gemm_tn_multi launch (8-wave kernel, then gemm_tn4)";
  timeout 100 tools/bin/gemm_lab tnmulti 32768 1 1 0; timeout 100 tools/bin/gemm_lab tnmulti 32768 1 1 1; } > gpurun_out/${TAG}_lab_nt.txt 2>&1
tail -4 gpurun_out/${TAG}_lab_nt.txt
# L2 requests / fabric-side traffic per launch of the lock-step kernels: gemm_nt3 (1003) against gemm_nt4 (2001)
timeout 400 bash tools/pmc_l2.sh 1003,2001 > gpurun_out/${TAG}_pmc_l2.txt 2>&1; tail -12 gpurun_out/${TAG}_pmc_l2.txt | cut -c1-200
timeout 300 bash tools/pmc_attn_run.sh > gpurun_out/${TAG}_pmc_attn.txt 2>&1; tail -2 gpurun_out/${TAG}_pmc_attn.txt | cut -c1-300
TAG=$TAG timeout 300 bash tools/prof_divae.sh > gpurun_out/${TAG}_divae_kernels.txt 2>&1; head -14 gpurun_out/${TAG}_divae_kernels.txt | cut -c1-200
