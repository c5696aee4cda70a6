#!/bin/bash
# Closing visit, third part: bench.py (default and driver flags) and the rocprofv3 kernel trace of the same headline command on ONE box.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/round5
mkdir -p $O/prof
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; cp gpurun_out/bench_detail_n1.json $O/bench_detail.json
echo "bench exit $? after $(( $(date +%s) - t0 )) s"
timeout -k 10 300 python bench.py --warmup 5 --steps 20 > $O/bench_driverflags.json 2>> $O/bench.err; cp gpurun_out/bench_detail_n1.json $O/bench_driverflags_detail.json
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_models; rm -rf $OUT/bench; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python $REPO/bench.py --no-cpu-baseline --no-extras --no-graph --no-tp > $OUT/bench.log 2>&1
python3 - <<PY
import csv, glob, json
rows = []
for f in glob.glob("$OUT/bench/**/*kernel_trace.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "gemm256p_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
timed = d[-400:-200] if len(d) >= 700 else d
out = {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extras --no-graph --no-tp",
       "kernel": rows[0]["Kernel_Name"] if rows else None, "launches": len(d),
       "us_mean_all": sum(d) / max(1, len(d)), "us_mean_timed_200": sum(timed) / max(1, len(timed)),
       "us_min_timed": min(timed) if timed else None, "us_max_timed": max(timed) if timed else None,
       "us_mean_first_100_warmup": sum(d[:100]) / max(1, len(d[:100]))}
json.dump(out, open("$OUT/bench_trace_summary.json", "w"), indent=1)
print(json.dumps(out))
PY
cd $REPO
for f in $(find gpurun_out/prof_models/bench -name "*kernel_stats.csv"); do cp $f $O/prof/; done
cp gpurun_out/prof_models/bench_trace_summary.json $O/prof/
echo "total $(( $(date +%s) - t0 )) s"
python -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['kernel_us'],{k:v for k,v in d['config'].items() if 'graph_ms' in k})"
