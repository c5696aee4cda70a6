#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r4c
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout 1200 python -m pytest tests/test_gpu_multi.py -m gpu -q --durations=8 > $O/pytest_multi.log 2>&1
echo "pytest_multi exit $? after $(( $(date +%s) - t0 )) s" | tee -a $O/pytest_multi.log
tail -60 $O/pytest_multi.log
