#!/bin/bash
# GPU-box session: (optional) tests, the headline bench, and a rocprofv3 kernel trace of the same command.
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -n "$PYTEST_ARGS" ]; then
  timeout 1200 python -m pytest $PYTEST_ARGS -m gpu -q -n 2 --timeout 600 --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -3; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu.log | head -20
fi
timeout ${BENCH_TIMEOUT:-900} python bench.py ${BENCH_ARGS:---steps 5 --warmup 2} > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench rc=$?"; tail -c 3000 gpurun_out/bench.json; tail -5 gpurun_out/bench.err
if [ -z "$NO_PROF" ]; then
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-profile > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof.err)
  echo "rocprof rc=$?"
  find gpurun_out/prof -name "*stats*" | head; 
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -25 "$f"
  # keep the summaries, drop the (large) per-dispatch trace
  find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
fi
