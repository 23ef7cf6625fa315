#!/bin/bash
# MFMA pipe utilisation of the GEMM kernels from rocprofv3 counters (stand-alone lab binary, counters only + kernel trace):
#   util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 1024 SIMDs)        (MI355X_MICROARCH.md: busy cycles = 32 per 32x32x16 MFMA)
cd "$(dirname "$0")/.."
export LD_LIBRARY_PATH=$PWD/ml-4m_amd/fourm/_lib:$LD_LIBRARY_PATH TMPDIR=/tmp
ROOT=$(pwd); mkdir -p gpurun_out; rm -rf gpurun_out/pmc_lab
for mode in tnmulti "nt 265"; do
  tag=$(echo $mode | cut -d' ' -f1)
  (cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_lab/$tag -o pmc -- $ROOT/tools/bin/gemm_lab $mode > /dev/null 2> $ROOT/gpurun_out/pmc_lab_$tag.err)
done
python - <<'PY'
import csv, glob, collections, re
# busy cycles per kernel from the counter file, durations from the kernel trace of the same run; utilisation quoted at the
# nominal 2.4 GHz (the chip runs below it under load: MI355X_MICROARCH.md "DVFS give-back"), 1024 SIMDs
for tag in ("tnmulti", "nt"):
    cf = glob.glob(f"gpurun_out/pmc_lab/{tag}/**/*counter_collection.csv", recursive=True)
    tf = glob.glob(f"gpurun_out/pmc_lab/{tag}/**/*kernel_trace.csv", recursive=True)
    if not cf or not tf:
        print(tag, "missing output"); continue
    short = lambda n: re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")[:60]
    busy, dur, n = collections.Counter(), collections.Counter(), collections.Counter()
    for row in csv.DictReader(open(cf[0], newline="")):
        if row["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            busy[short(row["Kernel_Name"])] += float(row["Counter_Value"])
    for row in csv.DictReader(open(tf[0], newline="")):
        k = short(row["Kernel_Name"])
        dur[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); n[k] += 1
    for k in busy:
        if busy[k] > 0 and dur[k] > 0:
            print(f"{tag:8s} {k:60s} launches {n[k]:5d}  avg {dur[k] / n[k] / 1e3:7.1f} us  MFMA busy {100 * busy[k] / (dur[k] * 2.4 * 1024):5.1f} % of 2.4 GHz x 1024 SIMDs")
PY
rm -rf gpurun_out/pmc_lab
