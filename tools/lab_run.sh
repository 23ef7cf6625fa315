#!/bin/bash
# scratch driver for one gpurun call: edit the command list, then  gpurun -- 'bash tools/lab_run.sh'
cd "$(dirname "$0")/.."
export LD_LIBRARY_PATH=$PWD/ml-4m_amd/fourm/_lib:$LD_LIBRARY_PATH TMPDIR=/tmp GEMM_LAB_NOWARM=1
ROOT=$(pwd); mkdir -p gpurun_out
for cfg in 266 267; do
rm -rf gpurun_out/pmc_nt
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_nt -o pmc -- $ROOT/tools/bin/gemm_lab nt $cfg > /dev/null 2>&1)
python - $cfg <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/pmc_nt/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f, newline="")) if r["Counter_Name"] == "FETCH_SIZE" and "gemm_nt_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
names = ["qkv N2304 K768", "N768 K768", "N2048 K768", "N1536 K768", "N768 K4096", "N768 K2304", "res N768 K2048", "res N768 K768", "swiglu 2x2048 K768"]
alg = [50 + 3.5, 50 + 1.2, 50 + 3.1, 50 + 2.4, 268 + 6.3, 151 + 3.5, 134 + 3.1 + 100, 50 + 1.2 + 100, 50 + 6.3]
per = len(rows) // 9
for i, (nm, al) in enumerate(zip(names, alg)):
    chunk = rows[i * per:(i + 1) * per]
    raw = sum(float(r["Counter_Value"]) for r in chunk) / len(chunk) / 1024
    print(f"cfg {sys.argv[1]} {nm:20s} launches {len(chunk):3d}  FETCH_SIZE raw {raw:7.1f} MB  x2 {2 * raw:7.1f} MB  algorithmic read {al:6.1f} MB  ratio(x2) {2 * raw / al:4.2f}")
PY
done
rm -rf gpurun_out/pmc_nt
