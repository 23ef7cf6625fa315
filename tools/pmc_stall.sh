#!/bin/bash
# L2 <-> fabric stall counters of the lock-step NT GEMM per shape, with and without its output stores (rocprofv3 --pmc, counters only +
# kernel trace; stand-alone lab binary).  usage: tools/pmc_stall.sh <cfg,cfg,...>     (1003 = default, 1043 = no stores)
cd "$(dirname "$0")/.."
export LD_LIBRARY_PATH=$PWD/ml-4m_amd/fourm/_lib:$LD_LIBRARY_PATH TMPDIR=/tmp GEMM_LAB_NOWARM=1
ROOT=$(pwd); mkdir -p gpurun_out; rm -rf gpurun_out/pmc_stall; mkdir -p gpurun_out/pmc_stall
CFGS=${1:-1003,1043}
i=0
for ctr in "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_CYCLE_sum" \
           "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum TCC_SRC_FIFO_FULL_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_stall/p$i -o pmc -- $ROOT/tools/bin/gemm_lab pmc $CFGS > $ROOT/gpurun_out/pmc_stall/order_$i.txt 2> $ROOT/gpurun_out/pmc_stall/err_$i.txt)
done
python - <<'PY'
import csv, glob, collections
order = [l.split() for l in open("gpurun_out/pmc_stall/order_1.txt") if " c" in l]
vals = collections.defaultdict(dict)
for p in (1, 2):
    cf = glob.glob(f"gpurun_out/pmc_stall/p{p}/**/*counter_collection.csv", recursive=True)
    if not cf: print("pass", p, "missing"); continue
    per = collections.defaultdict(lambda: collections.Counter())
    for row in csv.DictReader(open(cf[0], newline="")):
        if "gemm" in row["Kernel_Name"]:
            per[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
    ids = sorted(per)
    for g in range(len(ids) // 5):
        for k in per[ids[g * 5]]:
            vals[g][k] = sum(per[i][k] for i in ids[g * 5 + 1:g * 5 + 5]) / 4
for g, (name, cfg) in enumerate(order):
    v = vals.get(g, {})
    cyc = max(v.get("TCC_CYCLE_sum", 0), 1)
    f = lambda k: 100 * v.get(k, 0) / cyc
    print(f"{name:22s} {cfg:8s} TCC cycles {cyc / 1e6:8.1f} M | % of TCC cycles: busy {f('TCC_BUSY_sum'):5.1f}  EA write stall {f('TCC_EA0_WRREQ_STALL_sum'):5.1f}  "
          f"too many EA writes {f('TCC_TOO_MANY_EA_WRREQS_STALL_sum'):5.1f}  tag stall {f('TCC_TAG_STALL_sum'):5.1f}  DRAM wr credit {f('TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum'):5.1f}  "
          f"DRAM rd credit {f('TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum'):5.1f}  src fifo full {f('TCC_SRC_FIFO_FULL_sum'):5.1f}")
PY
