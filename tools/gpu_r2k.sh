#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_attention.py tests/test_gpu_plugin.py tests/test_gpu_models.py -m gpu -x -q 2>&1 | tail -3
python tools/attn_probe.py 2>&1 | tail -7
echo "--- bert"; python tools/model_bench.py bert 2>&1 | tail -1
