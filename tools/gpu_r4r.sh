#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r4r
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do
  timeout 600 python -m pytest tests/test_gpu_multi.py -m gpu -q -k "bench_tp_block or direct_transport" > $O/run_$rep.log 2>&1; echo "rep $rep exit $?" | tee -a $O/summary.txt
  grep -h "tp debug\|^FAILED" $O/run_$rep.log | cut -c1-400 | tee -a $O/summary.txt
done
