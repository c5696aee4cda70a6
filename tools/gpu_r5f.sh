#!/bin/bash
# Round 5, visit F: fp32 conv with the final split rule (per-layer table, fp32 tests) and the PMC pass of the 64 x 64-tile implicit GEMM.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r5f
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout -k 10 300 python -m pytest tests/test_gpu_nn.py tests/test_gpu_models.py -q -x -k "fp32 or f32 or reference_cpu_backend" > $O/pytest.log 2>&1
echo "pytest exit $? after $(( $(date +%s) - t0 )) s"; tail -3 $O/pytest.log | grep -v "version\|Hostname\|Librccl"
timeout -k 10 200 python tools/conv32_bench.py > $O/conv32.txt 2>&1; cut -c1-130 $O/conv32.txt | grep -v amdgpu
timeout -k 10 400 bash tools/profile_cmd.sh conv_igemm32 c32 -- python tools/run_conv32.py 128 28 128 3 1 1 > $O/prof_c32.log 2>&1; cp gpurun_out/prof_c32/summary.json $O/conv_igemm32_c128_28_pmc.json; cat $O/conv_igemm32_c128_28_pmc.json
echo "total $(( $(date +%s) - t0 )) s"
