#!/bin/bash
# Round 5: the library rebuilt from scratch (no cached objects) — smoke() and a cross-section of the parity suite.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout -k 10 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 100 python -m pytest tests/test_gpu_bench_routes.py tests/test_gpu_elementwise.py tests/test_gpu_movement.py -q -x 2>&1 | tail -2 | grep -v "version\|Hostname\|Librccl"
echo "total $(( $(date +%s) - t0 )) s"
