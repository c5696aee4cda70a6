#!/bin/bash
# HBM-side traffic of every memory-bound row of tools/membound_sweep.py: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in
# SEPARATE passes (never combined with other trace domains), FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note.
# -> gpurun_out/prof_membound/summary.json (copied to profiles/r0N_membound_pmc.json), stamped with the kernel sources' hash.
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_membound
rm -rf $OUT; mkdir -p $OUT
(cd $REPO && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python tools/membound_sweep.py --pmc-run > $OUT/fetch.log 2>&1)
(cd $REPO && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python tools/membound_sweep.py --pmc-run > $OUT/write.log 2>&1)
python3 - <<PY
import csv, glob, json
def disp(pat):
    rs = []
    for f in sorted(glob.glob(pat, recursive=True)):
        rs += list(csv.DictReader(open(f)))
    rs.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [r for r in rs if "irocm::" in r["Kernel_Name"]]
def segments(rs):
    segs, cur = [], None
    for r in rs:
        if "cast_kernel<float, signed char>" in r["Kernel_Name"] or ("cast" in r["Kernel_Name"] and "float" in r["Kernel_Name"] and "signed char" in r["Kernel_Name"]):
            if cur is not None: segs.append(cur)
            cur = []
        elif cur is not None:
            cur.append(r)
    return segs
listing = None
for line in open("$OUT/fetch.log"):
    if line.startswith("PMC_CASES "): listing = json.loads(line[10:])
fe, wr = segments(disp("$OUT/fetch/**/*counter_collection.csv")), segments(disp("$OUT/write/**/*counter_collection.csv"))
rows = {}
ok = listing is not None and len(fe) == len(listing["cases"]) == len(wr)
if ok:
    for c, f, w in zip(listing["cases"], fe, wr):
        fb = sum(float(r["Counter_Value"]) for r in f) * 1024 * 2 / c["launches"]
        wb = sum(float(r["Counter_Value"]) for r in w) * 1024 / c["launches"]
        rows[c["case"]] = {"shape": c["shape"], "algorithmic_bytes": c["algorithmic_bytes"], "fetch_bytes_x2": round(fb), "write_bytes": round(wb),
                           "traffic_over_algorithmic": round((fb + wb) / c["algorithmic_bytes"], 3),
                           "kernels": sorted({r["Kernel_Name"].split("(")[0][-70:] for r in f})}
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KB; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note); per launch, mean of 5; "
               "shapes below the 256 MiB Infinity Cache read less than their algorithmic bytes from HBM (cache hits are not HBM traffic)",
       "stamp": listing and listing["stamp"], "segments_found": [len(fe), len(wr)], "cases_expected": listing and len(listing["cases"]), "rows": rows}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps({k: v["traffic_over_algorithmic"] for k, v in rows.items()}))
print("ok" if ok else "SEGMENT MISMATCH", out["segments_found"], out["cases_expected"])
PY
