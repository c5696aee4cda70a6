#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2h
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider --deselect tests/test_gpu_matmul.py::test_matmul_16bit_variants > gpurun_out/r2h/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2h/pytest.log
tail -25 gpurun_out/r2h/pytest.log
timeout 300 python tools/model_bench.py resnet50 --batch 128 --iters 10 > gpurun_out/r2h/resnet.json 2> gpurun_out/r2h/resnet.err; tail -2 gpurun_out/r2h/resnet.json
