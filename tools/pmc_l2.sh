#!/bin/bash
# L2 behaviour of the lock-step NT GEMM per shape and tile order: TCC hit rate and fabric-side fetch (rocprofv3 --pmc, counters only +
# kernel trace; stand-alone lab binary).  usage: tools/pmc_l2.sh <cfg,cfg,...>
cd "$(dirname "$0")/.."
export LD_LIBRARY_PATH=$PWD/ml-4m_amd/fourm/_lib:$LD_LIBRARY_PATH TMPDIR=/tmp GEMM_LAB_NOWARM=1
ROOT=$(pwd); mkdir -p gpurun_out; rm -rf gpurun_out/pmc_l2; mkdir -p gpurun_out/pmc_l2
CFGS=${1:-1003}
i=0
for ctr in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_l2/p$i -o pmc -- $ROOT/tools/bin/gemm_lab pmc $CFGS > $ROOT/gpurun_out/pmc_l2/order_$i.txt 2> $ROOT/gpurun_out/pmc_l2/err_$i.txt)
done
python - <<'PY'
import csv, glob, collections
order = [l.split() for l in open("gpurun_out/pmc_l2/order_1.txt") if " c" in l]
vals = collections.defaultdict(dict)      # group -> counter -> per-launch mean
dur = {}
for p in (1, 2, 3):
    cf = glob.glob(f"gpurun_out/pmc_l2/p{p}/**/*counter_collection.csv", recursive=True)
    tf = glob.glob(f"gpurun_out/pmc_l2/p{p}/**/*kernel_trace.csv", recursive=True)
    if not cf: print("pass", p, "missing"); continue
    per = collections.defaultdict(lambda: collections.Counter())
    for row in csv.DictReader(open(cf[0], newline="")):
        if "gemm" in row["Kernel_Name"]:
            per[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
    ids = sorted(per)
    for g in range(len(ids) // 5):
        for k in per[ids[g * 5]]:
            vals[g][k] = sum(per[i][k] for i in ids[g * 5 + 1:g * 5 + 5]) / 4        # (first launch of a group: cold)
    if p == 1 and tf:
        rows = sorted((int(r["Dispatch_Id"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) for r in csv.DictReader(open(tf[0], newline="")) if "gemm" in r["Kernel_Name"])
        for g in range(len(rows) // 5):
            dur[g] = sum(d for _, d in rows[g * 5 + 1:g * 5 + 5]) / 4 / 1e3
for g, (name, cfg) in enumerate(order):
    v = vals.get(g, {})
    hit, miss = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    print(f"{name:22s} {cfg:8s} {dur.get(g, 0):7.1f} us (under counters)  L2 hit {100 * hit / max(hit + miss, 1):5.1f} %  req {(hit + miss) / 1e6:7.2f} M"
          f"  FETCH_SIZE x2 {2 * v.get('FETCH_SIZE', 0) / 1e3:8.1f} MB  WRITE_SIZE {v.get('WRITE_SIZE', 0) / 1e3:8.1f} MB")
PY
