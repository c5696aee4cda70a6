#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_attention.py tests/test_gpu_plugin.py tests/test_gpu_models.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -3
bash tools/profile_cmd.sh attention_kernel attn_bert -- python tools/attn_cmd.py
