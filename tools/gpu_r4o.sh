#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r4o
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_ab.sh "tools/conv_bench.py --variants=-1 --res --layers 3,7,13,19" "tools/model_bench.py resnet50" "tools/model_bench.py bert" > $O/ab_stdout.txt 2>&1
cp gpurun_out/ab/ab.txt $O/ab.txt
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest.log 2>&1
echo "pytest exit $?" | tee -a $O/pytest.log
tail -12 $O/pytest.log
