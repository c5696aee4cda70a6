#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_matmul.py tests/test_gpu_plugin.py tests/test_gpu_models.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -3
python tools/gemm_shapes.py 2>&1 | tee gpurun_out/gemm_shapes.txt | tail -10
echo "--- bert"; python tools/model_bench.py bert 2>&1 | tail -1
echo "--- resnet"; python tools/model_bench.py resnet50 2>&1 | tail -1
