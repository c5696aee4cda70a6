#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r4k
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo skip-pytest > $O/pytest.log
echo "pytest exit $?" | tee -a $O/pytest.log
tail -8 $O/pytest.log
timeout 200 python tools/attn_time.py 2>&1 | tee $O/attn_time.txt
timeout 200 python tools/run_stem.py 2>&1 | tee $O/stem_time.txt
timeout 300 python tools/model_bench.py resnet50 > $O/models.json 2> $O/models.err; timeout 300 python tools/model_bench.py bert >> $O/models.json 2>> $O/models.err
cat $O/models.json
