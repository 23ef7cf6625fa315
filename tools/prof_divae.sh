#!/bin/bash
# rocprofv3 kernel stats of UNet evaluations of the DiVAE detokenizer (bench.py --workload divae --pmc-worker = 2 evaluations at batch 8)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -rf gpurun_out/prof_divae
export TMPDIR=/tmp
ROOT=$(pwd)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_divae -o trace -- python $ROOT/bench.py --workload divae --pmc-worker > /dev/null 2> $ROOT/gpurun_out/prof_divae.err)
f=$(find gpurun_out/prof_divae -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${TAG:-r06}_divae_kernel_stats.csv
find gpurun_out/prof_divae -name "*kernel_trace.csv" -delete
find gpurun_out/prof_divae -name "*.db" -delete
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${TAG:-r06}_divae_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms (2 evaluations + setup):", tot/1e6)
for r in rows[:28]:
    print(f'{float(r["TotalDurationNs"])/1e6:8.3f} ms {int(r["Calls"]):5d} calls {float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][:110]}')
PY
