#!/bin/bash
# a short GPU visit: the tests named on the command line (pytest -k expression), default: everything
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/check
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q -x ${1:+-k "$1"} > $O/pytest.log 2>&1
echo "pytest exit $?" | tee -a $O/pytest.log
tail -15 $O/pytest.log
