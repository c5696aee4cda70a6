#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2d
export TMPDIR=/tmp
timeout 200 python tools/gemm_timeline.py --m 16384 --n 3072 --k 768 --tile 256 --wg 0,100,255 > gpurun_out/r2d/tl_ffn1.log 2>&1; cat gpurun_out/r2d/tl_ffn1.log
timeout 200 python tools/gemm_timeline.py --m 4096 --n 4096 --k 4096 --tile 256 --wg 0,100 > gpurun_out/r2d/tl_head.log 2>&1; head -c 6000 gpurun_out/r2d/tl_head.log
timeout 200 python tools/gemm_timeline.py --m 16384 --n 768 --k 768 --tile 192 --wg 0,100 > gpurun_out/r2d/tl_qkv.log 2>&1; cat gpurun_out/r2d/tl_qkv.log
