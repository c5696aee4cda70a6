#!/bin/bash
# rocprofv3 PMC passes of the headline GEMM (separate runs; never combined with --sys-trace etc.).
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_gemm
rm -rf $OUT; mkdir -p $OUT
V=${1:-2}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$name -o pmc -- python $REPO/tools/run_gemm.py 4096 $V 0 5 > $OUT/pmc_$name.log 2>&1
done
python3 - <<PY
import csv, glob, collections
tot={}
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        tot[k]=sum(v)/len(v); print(f"{k:36s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
if "GRBM_GUI_ACTIVE" in tot and "SQ_VALU_MFMA_BUSY_CYCLES" in tot:
    print("mfma_busy_frac", tot["SQ_VALU_MFMA_BUSY_CYCLES"]/1024/(tot["GRBM_GUI_ACTIVE"]/8))
PY
