#!/bin/bash
# Round 5, visit I: fp32 conv with inline-asm gather loads and hand-counted waits — parity (every fp32 conv / model test), per-layer table.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r5i
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout -k 10 300 python -m pytest tests/test_gpu_nn.py tests/test_gpu_models.py tests/test_gpu_frontend_exports.py -q -x -k "fp32 or f32 or reference_cpu_backend or export" > $O/pytest.log 2>&1
echo "pytest exit $? after $(( $(date +%s) - t0 )) s"; tail -3 $O/pytest.log | grep -v "version\|Hostname\|Librccl"
timeout -k 10 200 python tools/conv32_bench.py > $O/conv32.txt 2>&1; cut -c1-130 $O/conv32.txt | grep -v amdgpu
echo "total $(( $(date +%s) - t0 )) s"
