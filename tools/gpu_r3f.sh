#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r3f
mkdir -p $O
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_matmul.py tests/test_gpu_nn.py -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest exit $?"; tail -3 $O/pytest.log
for s in "2048 2048" "16384 3072"; do set -- $s; echo "== m $1 n $2"; timeout 100 python tools/gemm_timeline.py --m $1 --n $2 --k 768 --wg 0 2>&1 | grep -v amdgpu; done > $O/timeline.txt; cat $O/timeline.txt
timeout 200 python tools/gemm_shapes.py --dtype bf16 > $O/gemm_shapes_bf16.txt 2>&1; cat $O/gemm_shapes_bf16.txt | grep -v amdgpu
timeout 300 python tools/conv_bench.py --variants=-1   > $O/conv.txt 2>&1; grep -v amdgpu $O/conv.txt | cut -c1-150
timeout 200 python bench.py --no-tp --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-700
echo "total $(( $(date +%s) - t0 )) s"
