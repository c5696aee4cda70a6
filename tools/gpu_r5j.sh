#!/bin/bash
# Round 5, visit J: row kernels with a residency-sized persistent grid — parity and the sweep rows.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r5j
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout -k 10 300 python -m pytest tests/test_gpu_rowops.py -q -x > $O/pytest.log 2>&1
echo "pytest exit $? after $(( $(date +%s) - t0 )) s"; tail -3 $O/pytest.log | grep -v "version\|Hostname\|Librccl"
timeout -k 10 200 python tools/membound_sweep.py --only softmax,softmax_f32,layernorm,layernorm_f32,rmsnorm,add_rmsnorm,add_layernorm > $O/membound.txt 2>&1; grep -v amdgpu $O/membound.txt
echo "total $(( $(date +%s) - t0 )) s"
