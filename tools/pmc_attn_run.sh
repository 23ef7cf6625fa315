cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  n=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_attn2 -o $n -- python $R/tools/pmc_attn.py > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ['GRAFT_REPO_ROOT']
tot=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob(R+'/gpurun_out/pmc_attn2/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name']
        if 'attn_' not in k: continue
        key='bwd' if 'attn_bwd' in k else 'fwd'
        tot[key][row['Counter_Name']]+=float(row['Counter_Value'])
for key in tot:
    print(key, {c: f"{v/3:.3g}" for c,v in sorted(tot[key].items())})
PY
