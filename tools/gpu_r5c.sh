#!/bin/bash
# Round 5, visit C: whole GPU suite (shaped glue, fp32 pipeline, stem tile order), fp32 conv forms, stem + ResNet line.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r5c
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q --durations=5 > $O/pytest.log 2>&1
echo "pytest exit $? after $(( $(date +%s) - t0 )) s" | tee -a $O/pytest.log
tail -12 $O/pytest.log
timeout 400 python tools/conv32_bench.py --forms > $O/conv32.txt 2>&1; cat $O/conv32.txt | cut -c1-220
timeout 200 python tools/conv_bench.py --variants=-1 --layers 0 > $O/stem.txt 2>&1; cat $O/stem.txt | cut -c1-200
timeout 300 python tools/model_bench.py resnet50 > $O/resnet.txt 2>&1; tail -2 $O/resnet.txt | cut -c1-400
echo "total $(( $(date +%s) - t0 )) s"
