#!/bin/bash
# rocprofv3 PMC passes of an arbitrary command, summarised for kernels matching a name filter.
#   tools/profile_cmd.sh <name-filter> <out-tag> -- <command...>
# Separate passes per counter set (never combined with --sys-trace etc.); results under gpurun_out/prof_<tag>/.
FILTER=$1; TAG=$2; shift 3
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd $REPO && rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o pmc -- "$@" > $OUT/p$i.log 2>&1)
done
python3 - <<PY
import csv, glob, collections, json
tot = {}; dur = []
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "$FILTER" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        tot[k] = sum(v) / len(v)
for f in sorted(glob.glob("$OUT/p3/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "$FILTER" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {"filter": "$FILTER", "command": "$*", "pmc_per_dispatch_mean": tot, "dispatches": len(dur),
       "avg_us_under_profiler": sum(dur) / max(1, len(dur))}
if "GRBM_GUI_ACTIVE" in tot and "SQ_VALU_MFMA_BUSY_CYCLES" in tot:
    out["mfma_busy_frac"] = tot["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (tot["GRBM_GUI_ACTIVE"] / 8)
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    out["traffic_bytes_per_launch"] = tot["FETCH_SIZE"] * 1024 * 2 + tot["WRITE_SIZE"] * 1024
if "SQ_LDS_BANK_CONFLICT" in tot and "SQ_LDS_IDX_ACTIVE" in tot and tot["SQ_LDS_IDX_ACTIVE"]:
    out["lds_bank_conflict_frac"] = tot["SQ_LDS_BANK_CONFLICT"] / tot["SQ_LDS_IDX_ACTIVE"]
print(json.dumps(out, indent=1))
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
PY
