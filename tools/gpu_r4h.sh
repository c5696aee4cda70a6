#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r4h
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench exit $? after $(( $(date +%s) - t0 )) s"
timeout 300 python bench.py --warmup 5 --steps 20 --no-graph --no-tp --no-cpu-baseline --no-extras > $O/bench_driverflags.json 2>> $O/bench.err
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log
timeout 300 python tools/conv_bench.py --variants=-1 > $O/conv_layers.txt 2>&1
echo "total $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4h/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "roofline", "config") if k in d})
print("cpu_baseline", d.get("cpu_baseline"))
print("graph", json.dumps(d.get("graph_resnet50"))[:400])
print("tp", json.dumps(d.get("tp_block"))[:600])
ex = d.get("extras", {})
print("extras keys", list(ex.keys())[:12])
mb = ex.get("membound", {})
print("membound", str(mb)[:600])
PY
cut -c1-300 $O/bench_driverflags.json; tail -3 $O/smoke.log; tail -4 $O/conv_layers.txt; tail -5 $O/bench.err
