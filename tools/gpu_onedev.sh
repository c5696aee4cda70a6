#!/bin/bash
# the TP block of bench.py with N ranks on ONE device over the hand-written transport (timings of the transport's own cost only)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/onedev
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 IROCM_BENCH_ONE_DEVICE=1 INFINI_ROCM_COMM=direct
for w in 2 4; do
  timeout 300 python bench.py --gpus $w --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-extras --print-detail > $O/b$w.json 2> $O/b$w.err
  python - <<PY
import json
d=json.loads(open("$O/b$w.json").read().strip().splitlines()[-1])["tp_block"]
print("world $w:", {k:d.get(k) for k in ("allreduce_16MiB_ms","allreduce_busbw_GBs","ms_per_block","max_abs_diff_vs_unsharded")}, d.get("overlap"))
PY
done
