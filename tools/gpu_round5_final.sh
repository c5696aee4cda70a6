#!/bin/bash
# The closing GPU visit of round 5 (lean form of tools/gpu_round5.sh: counter passes only for kernels whose sources changed since the
# previous evidence visit — the memory-bound rows (nnops.hip) and the stem; the GEMM / tap GEMM / depthwise / decode-attention / fp32 conv
# counter files of this round stay). Every rocprofv3 step runs under `timeout -k`. Results under gpurun_out/round5/.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/round5
rm -rf $O; mkdir -p $O/prof
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=12 > $O/pytest.log 2>&1
echo "pytest exit $? after $(( $(date +%s) - t0 )) s" | tee -a $O/pytest.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke exit $?" | tee -a $O/smoke.log
timeout -k 10 300 bash tools/profile_membound.sh > $O/prof_membound.log 2>&1 && cp gpurun_out/prof_membound/summary.json $O/prof/membound_pmc.json && cp $O/prof/membound_pmc.json profiles/r05_membound_pmc.json
timeout -k 10 300 bash tools/profile_cmd.sh conv_stem_pool stem -- python tools/run_stem.py > $O/prof_stem.log 2>&1 && cp gpurun_out/prof_stem/summary.json $O/prof/conv_stem_pool_pmc.json
echo "pmc done after $(( $(date +%s) - t0 )) s"
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; cp gpurun_out/bench_detail_n1.json $O/bench_detail.json
echo "bench exit $? after $(( $(date +%s) - t0 )) s"
timeout -k 10 300 python bench.py --warmup 5 --steps 20 > $O/bench_driverflags.json 2>> $O/bench.err; cp gpurun_out/bench_detail_n1.json $O/bench_driverflags_detail.json
for m in "resnet50 --tune" "bert --tune" "bert --decomposed" "llama"; do
  timeout -k 10 240 python tools/model_bench.py $m >> $O/models.json 2>> $O/models.err
done
echo "models done after $(( $(date +%s) - t0 )) s"
timeout -k 10 300 python tools/membound_sweep.py --json $O/membound.json > $O/membound.txt 2>&1
timeout -k 10 300 python tools/conv_bench.py --variants=-1,2,7 > $O/conv_layers.txt 2>&1
timeout -k 10 200 python tools/dwconv_bench.py > $O/dwconv_layers.txt 2>&1
timeout -k 10 200 python tools/conv32_bench.py --forms > $O/conv32_layers.txt 2>&1
(for a in "--bh 32 --n 4096" "--bh 32 --n 32768" "--bh 256 --n 2048" "--bh 8 --n 8192 --d 256 --dtype bf16" "--bh 32 --n 4096 --dtype f32"; do timeout -k 10 100 python tools/kvcache_bench.py $a --splits 0,-1; done) > $O/kvcache.txt 2>&1
python tools/run_stem.py > $O/stem.txt 2>&1; IROCM_STEM_LINEAR=1 python tools/run_stem.py 2>&1 | sed 's/^/round-4 tile order: /' >> $O/stem.txt
INFINI_ROCM_FUSION_LOG=1 timeout -k 10 200 python tools/model_bench.py resnet50 --iters 1 2> $O/resnet50_plan_log.txt > /dev/null
echo "sweeps done after $(( $(date +%s) - t0 )) s"
rm -rf gpurun_out/prof_models; timeout -k 10 500 bash tools/profile_models.sh > $O/prof_models.log 2>&1
for f in $(find gpurun_out/prof_models -name "*kernel_stats.csv"); do cp $f $O/prof/; done
cp gpurun_out/prof_models/bench_trace_summary.json $O/prof/ 2>/dev/null
echo "total $(( $(date +%s) - t0 )) s"
tail -4 $O/pytest.log 2>/dev/null | grep -v "version\|Hostname\|Librccl"; tail -2 $O/smoke.log; cut -c1-400 $O/bench.json; echo; cut -c1-330 $O/models.json; tail -2 $O/conv_layers.txt; cat $O/stem.txt; cat $O/prof/conv_stem_pool_pmc.json | tail -8
