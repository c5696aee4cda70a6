#!/bin/bash
# rocprofv3 kernel trace + stats of the graph-level benchmarks and of bench.py; CSV summaries under gpurun_out/prof_models
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_models
mkdir -p $OUT
for m in "resnet50 --batch 128 --iters 3" "bert --batch 32 --seq 512 --iters 3" "bert --batch 32 --seq 512 --iters 3 --decomposed" "llama --iters 3"; do
  name=$(echo $m | cut -d' ' -f1)$(echo $m | grep -q decomposed && echo _decomposed)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o $name -- python $REPO/tools/model_bench.py $m > $OUT/$name.log 2>&1
done
# the SAME headline command bench.py runs by default (300 warm-up + 200 timed launches); the secondary sections are switched off
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python $REPO/bench.py --no-cpu-baseline --no-extras --no-graph --no-tp > $OUT/bench.log 2>&1
python3 - <<PY
import csv, glob, json
rows = []
for f in glob.glob("$OUT/bench/**/*kernel_trace.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "gemm128w_kernel" in r["Kernel_Name"] or "gemm256p_kernel" in r["Kernel_Name"]]
# the headline launches ONE GEMM kernel (whichever the heuristic picks): keep the name with the most launches
names = {}
for r in rows:
    names[r["Kernel_Name"]] = names.get(r["Kernel_Name"], 0) + 1
top = max(names, key=names.get) if names else None
rows = [r for r in rows if r["Kernel_Name"] == top]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
# launch order: 300 warm-up, 200 timed, then whatever later sections add (none with the flags above)
# launch order: 2 + 20 cold launches, the time-based pre-warm (a multiple of 16), 300 warm-up, 200 timed, then the per-launch pass
timed = d[-400:-200] if len(d) >= 700 else d
out = {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extras --no-graph --no-tp",
       "kernel": rows[0]["Kernel_Name"] if rows else None, "launches": len(d),
       "us_mean_all": sum(d) / max(1, len(d)), "us_mean_timed_200": sum(timed) / max(1, len(timed)),
       "us_min_timed": min(timed) if timed else None, "us_max_timed": max(timed) if timed else None,
       "us_mean_first_100_warmup": sum(d[:100]) / max(1, len(d[:100]))}
json.dump(out, open("$OUT/bench_trace_summary.json", "w"), indent=1)
print(json.dumps(out))
PY
for f in $(find $OUT -name "*kernel_stats.csv"); do echo "== $f"; head -14 $f | cut -c1-230; done
