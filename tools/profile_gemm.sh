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
import json
sys_path = "$REPO"
import sys
sys.path.insert(0, sys_path)
try:
    from infinitensor_amd import ops
    vname = ops.matmul_variants()[int("$V")]
except Exception:
    vname = None
kern = sorted({r["Kernel_Name"] for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "gemm" in r.get("Kernel_Name", "")})
sys.path.insert(0, "$REPO/tools")
import source_stamps
out = {"stamp": source_stamps.gemm_stamp(), "command": "rocprofv3 --kernel-trace --pmc <one set per pass> -- python tools/run_gemm.py 4096 $V 0 5 (bf16 NN 4096^3)",
       "variant": int("$V"), "variant_name": vname, "kernel": kern[0] if kern else None, "pmc_per_dispatch_mean": tot}
if "GRBM_GUI_ACTIVE" in tot and "SQ_VALU_MFMA_BUSY_CYCLES" in tot:
    # SQ counters are summed over 1024 SIMDs (256 CU x 4), GRBM_GUI_ACTIVE over 8 XCDs
    out["mfma_busy_frac"] = tot["SQ_VALU_MFMA_BUSY_CYCLES"]/1024/(tot["GRBM_GUI_ACTIVE"]/8)
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    # KB per dispatch; gfx950 correction from MI355X_MICROARCH.md (HBM section): FETCH_SIZE x2 for 16 B/lane reads
    out["traffic_bytes_per_launch"] = tot["FETCH_SIZE"]*1024*2 + tot["WRITE_SIZE"]*1024
    out["algorithmic_bytes_per_launch"] = 3*4096*4096*2
print(json.dumps(out))
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
PY
