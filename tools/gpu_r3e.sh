#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r3e
mkdir -p $O
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_nn.py tests/test_gpu_plugin.py tests/test_gpu_models.py tests/test_gpu_fuzz.py -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest exit $?"; tail -5 $O/pytest.log
timeout 300 python tools/conv_bench.py --variants=-1,2,5   > $O/conv_pw.txt 2>&1; grep -v amdgpu $O/conv_pw.txt | tail -4
timeout 300 python tools/conv_fusion_fuzz.py 40 > $O/conv_fuzz.txt 2>&1; tail -3 $O/conv_fuzz.txt
for m in "resnet50" "resnet50 --tune" "resnet50 --idealised"; do timeout 300 python tools/model_bench.py $m 2>/dev/null | tail -1 | tee -a $O/resnet50.txt; done
echo "total $(( $(date +%s) - t0 )) s"
