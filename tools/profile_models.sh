#!/bin/bash
# rocprofv3 kernel trace + stats of the graph-level benchmarks and of bench.py; CSV summaries under gpurun_out/prof_models
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_models
mkdir -p $OUT
for m in "resnet50 --batch 128 --iters 3" "bert --batch 32 --seq 512 --iters 3"; do
  name=$(echo $m | cut -d' ' -f1)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o $name -- python $REPO/tools/model_bench.py $m > $OUT/$name.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python $REPO/bench.py --no-cpu-baseline --no-extras --no-graph --no-tp --steps 50 --warmup 5 > $OUT/bench.log 2>&1
for f in $(find $OUT -name "*kernel_stats.csv"); do echo "== $f"; head -14 $f | cut -c1-230; done
