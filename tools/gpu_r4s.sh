#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r4s
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2 3 4 5; do
  timeout 300 python -m pytest tests/test_gpu_multi.py -m gpu -q -k "direct_transport and 8" > $O/run_$rep.log 2>&1; echo "rep $rep exit $?" | tee -a $O/summary.txt
  grep -h "ring round\|multi-piece\|AssertionError:" $O/run_$rep.log | cut -c1-300 | tee -a $O/summary.txt
done
