#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r4q
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
export IROCM_BENCH_TP_DEBUG=1 IROCM_BENCH_ONE_DEVICE=1 INFINI_ROCM_COMM=direct
for rep in 1 2 3 4 5 6; do
  for w in 2 4; do
    timeout 300 python bench.py --gpus $w --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-extras > $O/b_${w}_$rep.json 2> $O/b_${w}_$rep.err
    echo "world $w rep $rep rc $?: $(grep 'tp debug' $O/b_${w}_$rep.err | head -1) | $(python -c "
import json,sys
try:
    d=json.loads(open('$O/b_${w}_$rep.json').read().strip().splitlines()[-1]); print(d['tp_block']['max_abs_diff_vs_unsharded'])
except Exception as e: print('noline',e)
")" | tee -a $O/summary.txt
  done
done
