#!/bin/bash
# round-2 GPU call A: full GPU suite + GEMM variant sweep + MFMA ceiling
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -x --deselect tests/test_gpu_matmul.py::test_matmul_16bit_variants -p no:cacheprovider > gpurun_out/r2a/pytest_main.log 2>&1
echo "pytest main rc=$?" >> gpurun_out/r2a/pytest_main.log
tail -30 gpurun_out/r2a/pytest_main.log
timeout 600 python -m pytest tests/test_gpu_matmul.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "test_matmul_16bit_variants" > gpurun_out/r2a/pytest_variants.log 2>&1
tail -5 gpurun_out/r2a/pytest_variants.log
timeout 120 python tools/mfma_ceiling.py > gpurun_out/r2a/mfma_ceiling.json 2>&1; cat gpurun_out/r2a/mfma_ceiling.json
timeout 300 python tools/gemm_shapes.py --dtype bf16 --variants=-1,2,4,5,6,7,8,9 --iters 50 > gpurun_out/r2a/gemm_bf16.log 2>&1; cat gpurun_out/r2a/gemm_bf16.log
timeout 300 python tools/gemm_shapes.py --dtype f16 --variants=-1,1,2,3,4,5,6 --iters 50 > gpurun_out/r2a/gemm_f16.log 2>&1; cat gpurun_out/r2a/gemm_f16.log
