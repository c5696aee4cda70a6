#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2i
export TMPDIR=/tmp
IROCM_ATTN_NT=2 timeout 600 python -m pytest tests/test_gpu_attention.py -m gpu -q --maxfail=12 -p no:cacheprovider > gpurun_out/r2i/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2i/pytest.log
tail -5 gpurun_out/r2i/pytest.log
for a in 0 2; do echo "== NT env $a"; IROCM_ATTN_NT=$a timeout 200 python tools/attn_probe.py 2>&1 | grep -E "bh384|bh96"; done > gpurun_out/r2i/attn.log 2>&1; cat gpurun_out/r2i/attn.log
