#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r4p
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2 3; do
  timeout 600 python -m pytest tests/test_gpu_multi.py -m gpu -q -x -k "bench_tp_block" > $O/new_$rep.log 2>&1; echo "new rep $rep exit $?" | tee -a $O/summary.txt
  grep -o "max_abs_diff_vs_unsharded': [a-z0-9.e-]*" $O/new_$rep.log | tee -a $O/summary.txt
done
export INFINI_ROCM_LIB=$REPO/infinitensor_amd/lib/ab/base.so
for rep in 1 2; do
  timeout 600 python -m pytest tests/test_gpu_multi.py -m gpu -q -x -k "bench_tp_block" > $O/base_$rep.log 2>&1; echo "base rep $rep exit $?" | tee -a $O/summary.txt
done
