#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r3g
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in "resnet50 --batch 128 --iters 3" "bert --batch 32 --seq 512 --iters 3"; do
  name=$(echo $m | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o $name -- python $REPO/tools/model_bench.py $m > $O/$name.log 2>&1
  tail -1 $O/$name.log | cut -c1-400
  f=$(find $O/$name -name "*kernel_stats.csv" | head -1); cp $f $O/${name}_kernel_stats.csv; head -24 $f | cut -c1-200
done
