#!/bin/bash
# Round 5, last visit: the whole GPU suite and smoke() on the final tree.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r5z
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1
echo "pytest exit $? after $(( $(date +%s) - t0 )) s" | tee -a $O/pytest.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" | tee -a $O/smoke.log
tail -6 $O/pytest.log | grep -v "version\|Hostname\|Librccl"; tail -2 $O/smoke.log
echo "total $(( $(date +%s) - t0 )) s"
