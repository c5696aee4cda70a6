#!/bin/bash
# attention NT=2 default + mask strip; Gelu as a compile-time GEMM epilogue; BERT with / without the Gelu fusion
mkdir -p gpurun_out
python -m pytest tests/test_gpu_attention.py tests/test_gpu_matmul.py -m gpu -x -q 2>&1 | tail -3
python tools/attn_probe.py 2>&1 | tail -7
echo "--- bert"; python tools/model_bench.py bert 2>&1 | tail -1
echo "--- bert fuse gelu"; INFINI_ROCM_FUSE_GELU=1 python tools/model_bench.py bert 2>&1 | tail -1
