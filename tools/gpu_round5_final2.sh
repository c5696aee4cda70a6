#!/bin/bash
# Closing visit, second part: the figures that the tile-width rule of the pointwise conv mode moved (bench lines, ResNet-50 lines and
# kernel trace, per-layer conv tables, the memory-bound sweep after the resident-grid fixes). Results into gpurun_out/round5/ (merged
# over the first part's files; tools/collect_profiles.sh copies them).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/round5
mkdir -p $O/prof
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; cp gpurun_out/bench_detail_n1.json $O/bench_detail.json
echo "bench exit $? after $(( $(date +%s) - t0 )) s"
timeout -k 10 300 python bench.py --warmup 5 --steps 20 > $O/bench_driverflags.json 2>> $O/bench.err; cp gpurun_out/bench_detail_n1.json $O/bench_driverflags_detail.json
rm -f $O/models.json
for m in "resnet50 --tune" "bert --tune" "bert --decomposed" "llama"; do
  timeout -k 10 240 python tools/model_bench.py $m >> $O/models.json 2>> $O/models.err
done
timeout -k 10 300 python tools/membound_sweep.py --json $O/membound.json > $O/membound.txt 2>&1
timeout -k 10 300 python tools/conv_bench.py --variants=-1,2,7 > $O/conv_layers.txt 2>&1
timeout -k 10 200 python tools/conv_bench.py --variants=-1,2 --res --layers 3,7,13,19 > $O/conv_layers_residual.txt 2>&1
INFINI_ROCM_FUSION_LOG=1 timeout -k 10 200 python tools/model_bench.py resnet50 --iters 1 2> $O/resnet50_plan_log.txt > /dev/null
echo "sweeps done after $(( $(date +%s) - t0 )) s"
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_models; rm -rf $OUT/resnet50 $OUT/bench; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/resnet50 -o resnet50 -- python $REPO/tools/model_bench.py resnet50 --batch 128 --iters 3 > $OUT/resnet50.log 2>&1
cd $REPO
for f in $(find gpurun_out/prof_models/resnet50 -name "*kernel_stats.csv"); do cp $f $O/prof/; done
echo "total $(( $(date +%s) - t0 )) s"
cut -c1-300 $O/bench.json; echo; grep -o '"resnet50[^,]*' $O/bench.json | head -4; grep -o '"resnet50[^,]*' $O/bench_driverflags.json | head -2; cut -c1-200 $O/models.json; tail -2 $O/conv_layers.txt; tail -1 $O/conv_layers_residual.txt
