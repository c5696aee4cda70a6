#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2i
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_nn.py tests/test_gpu_plugin.py tests/test_gpu_models.py tests/test_gpu_matmul.py -m gpu -q --maxfail=12 --deselect tests/test_gpu_matmul.py::test_matmul_16bit_variants -p no:cacheprovider > gpurun_out/r2i/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2i/pytest.log
tail -5 gpurun_out/r2i/pytest.log
