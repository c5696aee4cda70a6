#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/timeline
rm -rf $O; mkdir -p $O
timeout 200 python tools/gemm_timeline.py --m 16384 --n 3072 --k 768 --tile 256 --wg 0 2>&1 | grep -v amdgpu.ids | tee $O/timeline_ffn1.txt
timeout 200 python tools/gemm_timeline.py --m 16384 --n 2304 --k 768 --tile 192 --wg 0 2>&1 | grep -v amdgpu.ids | tee $O/timeline_qkv.txt
