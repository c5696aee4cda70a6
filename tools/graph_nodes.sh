#!/bin/bash
# Which kernels does ONE hipGraph replay of a model graph launch, in order (rocprofv3 kernel trace of tools/model_bench.py)? Prints the last
# replay's sequence with every __amd_rocclr_* helper (memcpy / memset nodes) marked.   usage: tools/graph_nodes.sh MODEL [args...]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
M=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/graph_nodes/$M
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $REPO/tools/model_bench.py $M "$@" > $OUT/run.log 2>&1
tail -1 $OUT/run.log | cut -c1-400
python3 - <<PY
import csv, glob, re
from collections import Counter
rows = []
for f in sorted(set(glob.glob("$OUT/**/*kernel_trace.csv", recursive=True))):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\\(.*", "", n)
    return n.replace("void irocm::", "").replace("irocm::", "")[:64]
names = [short(r["Kernel_Name"]) for r in rows]
end = len(names) - 1  # (the very last kernel is the output's copy-out: leave it outside)
period = None
for pp in range(5, 800):
    if end - 3 * pp < 0:
        break
    if names[end - pp:end] == names[end - 2 * pp:end - pp] == names[end - 3 * pp:end - 2 * pp]:
        period = pp
        break
print(f"{len(rows)} kernels traced; one replay = {period} launches")
if period:
    seq = range(end - period, end)
    c = Counter(names[k] for k in seq)
    print("helper (memcpy / memset) nodes per replay:", {k: v for k, v in c.items() if "rocclr" in k} or "none")
    for k in seq:
        r = rows[k]; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if period <= 40 or "rocclr" in names[k]:
            print(f"  {d:8.1f} us  {names[k]}" + (f"   (after {names[k-1]} | before {names[k+1]})" if "rocclr" in names[k] and period > 40 else ""))
PY
