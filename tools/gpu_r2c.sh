#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2c
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_matmul.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "persistent or headline or head_split" > gpurun_out/r2c/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2c/pytest.log
tail -8 gpurun_out/r2c/pytest.log
timeout 300 python tools/gemm_shapes.py --dtype bf16 --variants=-1,2,4,5,6,7,8,9 --iters 50 > gpurun_out/r2c/gemm_bf16.log 2>&1; cat gpurun_out/r2c/gemm_bf16.log
timeout 300 python tools/gemm_shapes.py --dtype f16 --variants=-1,2,4,5,6,7,8,9 --iters 50 > gpurun_out/r2c/gemm_f16.log 2>&1; cat gpurun_out/r2c/gemm_f16.log
