#!/bin/bash
# rocprofv3 of the headline GEMM: kernel trace + stats, then PMC passes (separate runs; never combined
# with --sys-trace etc.). Writes under gpurun_out/prof_gemm/.
set -x
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_gemm
mkdir -p $OUT
V=${1:-2}
rocprofv3 --kernel-trace --stats -d $OUT/trace -o gemm -- python $REPO/tools/run_gemm.py 4096 $V 0 20 > $OUT/trace.log 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_$name -o pmc -- python $REPO/tools/run_gemm.py 4096 $V 0 5 > $OUT/pmc_$name.log 2>&1
done
find $OUT -name "*.csv" | head -50
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:40s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
for f in sorted(glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)):
    print(open(f).read()[:1500])
PY
