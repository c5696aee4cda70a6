#!/bin/bash
# rocprofv3 evidence for the HBM-bound rows: FETCH_SIZE / WRITE_SIZE (separate passes) + kernel durations of the softmax
# and LayerNorm kernels at the SURVEY 8d shapes -> gpurun_out/prof_rowops/summary.json (copied to profiles/ by hand).
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_rowops
rm -rf $OUT; mkdir -p $OUT
(cd $REPO && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python tools/rowops_cmd.py > $OUT/trace.log 2>&1)
(cd $REPO && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python tools/rowops_cmd.py > $OUT/fetch.log 2>&1)
(cd $REPO && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python tools/rowops_cmd.py > $OUT/write.log 2>&1)
python3 - <<PY
import csv, glob, json, collections
def rows(pat):
    out = []
    for f in sorted(glob.glob(pat, recursive=True)):
        out += list(csv.DictReader(open(f)))
    return out
# dispatch order is the same in every pass: key = (kernel name, occurrence index)
def keyed(rs, namecol="Kernel_Name"):
    seen = collections.Counter(); out = {}
    for r in rs:
        n = r[namecol]
        if "softmax" not in n and "norm_rows" not in n: continue
        out[(n, seen[n])] = r; seen[n] += 1
    return out
tr = keyed(sorted(rows("$OUT/trace/**/*kernel_trace.csv"), key=lambda r: int(r["Start_Timestamp"])))
fe = keyed(sorted(rows("$OUT/fetch/**/*counter_collection.csv"), key=lambda r: int(r["Dispatch_Id"])))
wr = keyed(sorted(rows("$OUT/write/**/*counter_collection.csv"), key=lambda r: int(r["Dispatch_Id"])))
# the command launches, per dtype: 6 x softmax 196608x512, 6 x LN 16384x768, 6 x LN 262144x768
names = collections.OrderedDict()
for (n, i) in tr: names.setdefault(n, []).append(i)
summary = []
for n, idx in names.items():
    esz = 2 if "__half" in n else 4
    groups = [idx[j:j + 6] for j in range(0, len(idx), 6)]
    for gi, g in enumerate(groups):
        if "softmax" in n: shape, numel = "196608x512", 196608 * 512
        else: shape, numel = (("16384x768", 16384 * 768) if gi == 0 else ("262144x768", 262144 * 768))
        use = g[2:]  # skip the first two (cold) launches
        us = sum((int(tr[(n, i)]["End_Timestamp"]) - int(tr[(n, i)]["Start_Timestamp"])) / 1e3 for i in use) / len(use)
        f = sum(float(fe[(n, i)]["Counter_Value"]) for i in use if (n, i) in fe) / len(use) * 1024 * 2   # KB, x2: gfx950 note
        w = sum(float(wr[(n, i)]["Counter_Value"]) for i in use if (n, i) in wr) / len(use) * 1024
        alg = 2 * numel * esz
        summary.append({"kernel": n.split("(")[0][-60:], "dtype": "f16" if esz == 2 else "f32", "shape": shape, "avg_us_under_profiler": round(us, 2),
                        "algorithmic_bytes": alg, "fetch_bytes_x2": round(f), "write_bytes": round(w),
                        "traffic_over_algorithmic": round((f + w) / alg, 3), "GBs_algorithmic": round(alg / us / 1e3, 1),
                        "frac_hbm_peak_8TBs": round(alg / us / 1e3 / 8000, 4)})
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KB units; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note), durations from the --kernel-trace pass; mean of launches 3..6", "rows": summary}, open("$OUT/summary.json", "w"), indent=1)
for s in summary: print(s)
PY
