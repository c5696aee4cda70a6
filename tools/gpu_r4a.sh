#!/bin/bash
# round 4, visit A: IPC feasibility probe, the new bench-size route tests + advisor tests, then the whole GPU suite.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r4a
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
(hipcc --offload-arch=gfx950 -O2 -w tools/probes/ipc_probe.hip -o /tmp/ipc_probe && timeout 60 /tmp/ipc_probe) > $O/ipc_probe.txt 2>&1
echo "ipc_probe exit $?" >> $O/ipc_probe.txt
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_bench_routes.py tests/test_gpu_rowops.py tests/test_gpu_plugin.py -m gpu -q -x --durations=8 > $O/pytest_new.log 2>&1
echo "pytest_new exit $? after $(( $(date +%s) - t0 )) s" | tee -a $O/pytest_new.log
timeout 900 python -m pytest tests -m gpu -q --durations=8 --deselect tests/test_gpu_bench_routes.py --deselect tests/test_gpu_rowops.py --deselect tests/test_gpu_plugin.py > $O/pytest_rest.log 2>&1
echo "pytest_rest exit $? after $(( $(date +%s) - t0 )) s" | tee -a $O/pytest_rest.log
cat $O/ipc_probe.txt; tail -25 $O/pytest_new.log; tail -8 $O/pytest_rest.log
