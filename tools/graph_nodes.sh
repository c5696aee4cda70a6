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
rows = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(.*", "", n); n = n.replace("void irocm::", "").replace("irocm::", "")
    return n[:70]
names = [short(r["Kernel_Name"]) for r in rows]
# a replay = the sequence between two occurrences of the graph's first kernel; take the last full one
first = None
for cand in names[::-1]:
    first = cand; break
# find the period: the last kernel name of the stream ends a replay; look for the previous identical tail
last = len(names) - 1
seq = []
i = last
# walk back until we see the same kernel as names[last] again (the previous replay's end)
j = last - 1
while j > 0 and not (names[j] == names[last] and j < last - 3): j -= 1
seq = list(range(j + 1, last + 1))
print(f"{len(rows)} kernels traced; last replay = {len(seq)} launches")
helpers = 0
for k in seq:
    r = rows[k]; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    mark = "  <== helper" if "rocclr" in names[k] else ""
    helpers += bool(mark)
    print(f"  {d:8.1f} us  {names[k]}{mark}")
print(f"helpers in the replay: {helpers}")
PY
