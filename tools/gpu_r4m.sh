#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r4m
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_matmul.py tests/test_gpu_bench_routes.py tests/test_gpu_fuzz.py tests/test_gpu_nn.py tests/test_gpu_models.py tests/test_gpu_plugin.py -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest exit $?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
bash tools/gpu_ab.sh "tools/gemm_shapes.py --variants -1 --dtype f16" "tools/conv_bench.py --variants=-1" "tools/model_bench.py resnet50" "tools/model_bench.py bert" > $O/ab_stdout.txt 2>&1
cp gpurun_out/ab/ab.txt $O/ab.txt
