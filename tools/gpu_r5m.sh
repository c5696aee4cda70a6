#!/bin/bash
# Round 5, visit M: strided 3 x 3 layers with 128 filters on the tap GEMM — route tests, tap tests, model tests, the layer's timing.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r5m
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout -k 10 400 python -m pytest tests/test_gpu_bench_routes.py tests/test_gpu_models.py tests/test_gpu_nn.py -q -x -k "route or tap or resnet or conv" > $O/pytest.log 2>&1
echo "pytest exit $? after $(( $(date +%s) - t0 )) s"; tail -3 $O/pytest.log | grep -v "version\|Hostname\|Librccl"
timeout -k 10 120 python tools/conv_bench.py --variants=-1,2 --layers 6,12 2>&1 | grep "^x" | cut -c1-190
echo "total $(( $(date +%s) - t0 )) s"
