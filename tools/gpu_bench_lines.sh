#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/round4
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1; do
  timeout 600 python bench.py > $O/bench_$rep.json 2> $O/bench_$rep.err; echo "bench rep $rep rc $?"
done
cp $O/bench_1.json $O/bench.json
timeout 300 python bench.py --warmup 5 --steps 20 --no-graph --no-tp --no-cpu-baseline --no-extras > $O/bench_driverflags.json 2>> $O/bench.err
cut -c1-300 $O/bench.json; echo; cut -c1-200 $O/bench_driverflags.json
