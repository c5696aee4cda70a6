#!/bin/bash
mkdir -p gpurun_out
INFINI_ROCM_FUSION_LOG=1 python tools/model_bench.py resnet50 --iters 1 2> gpurun_out/fusion_log_all.txt | tail -1
grep "^\[fusion\]" gpurun_out/fusion_log_all.txt | awk '!seen[$0]++' > gpurun_out/fusion_log.txt; rm gpurun_out/fusion_log_all.txt
