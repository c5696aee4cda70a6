#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_runtime.py tests/test_gpu_multi.py tests/test_gpu_plugin.py -m gpu -q --maxfail=30 -p no:cacheprovider > gpurun_out/r2b/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2b/pytest.log
tail -40 gpurun_out/r2b/pytest.log
timeout 300 python tools/gemm_shapes.py --dtype bf16 --variants=-1,2,4,5,6,7,8,9 --iters 50 > gpurun_out/r2b/gemm_bf16.log 2>&1; cat gpurun_out/r2b/gemm_bf16.log
timeout 300 python tools/gemm_shapes.py --dtype f16 --variants=-1,1,2,3,4,5,6 --iters 50 > gpurun_out/r2b/gemm_f16.log 2>&1; cat gpurun_out/r2b/gemm_f16.log
