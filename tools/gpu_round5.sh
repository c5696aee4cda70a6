#!/bin/bash
# One GPU visit of round 5: parity suite, smoke(), PMC passes (headline GEMM, memory-bound sweep, the stem, the tap GEMM, the depthwise and
# fp32 conv kernels, decode attention), bench.py (default flags and the driver's), model graphs, kernel traces, the per-layer sweeps.
# Everything lands under gpurun_out/round5/ (copied to profiles/r05_* afterwards). NO_PYTEST=1 / NO_PROFILE=1 skip those parts.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/round5
rm -rf $O; mkdir -p $O/prof
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
if [ -z "$NO_PYTEST" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/pytest.log 2>&1
  echo "pytest exit $? after $(( $(date +%s) - t0 )) s" | tee -a $O/pytest.log
fi
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke exit $?" | tee -a $O/smoke.log
if [ -z "$NO_PROFILE" ]; then
  # counters FIRST: bench.py below then finds counter files that carry the current kernels' stamp
  bash tools/profile_membound.sh > $O/prof_membound.log 2>&1; cp gpurun_out/prof_membound/summary.json $O/prof/membound_pmc.json
  bash tools/profile_gemm.sh 4 > $O/prof_gemm.log 2>&1; cp gpurun_out/prof_gemm/summary.json $O/prof/gemm256p_pmc.json
  cp $O/prof/membound_pmc.json profiles/r05_membound_pmc.json; cp $O/prof/gemm256p_pmc.json profiles/r05_gemm256p_pmc.json
  bash tools/profile_cmd.sh conv_stem_pool stem -- python tools/run_stem.py > $O/prof_stem.log 2>&1; cp gpurun_out/prof_stem/summary.json $O/prof/conv_stem_pool_pmc.json
  bash tools/profile_cmd.sh "4, false, 3>" tap7 -- python tools/conv_bench.py --variants=7 --layers 22 --iters 5 > $O/prof_tap7.log 2>&1; cp gpurun_out/prof_tap7/summary.json $O/prof/conv_tap_splitk_c512_7_pmc.json
  bash tools/profile_cmd.sh conv_dw dw2 -- python tools/dwconv_bench.py --layers 2 > $O/prof_dw.log 2>&1; cp gpurun_out/prof_dw2/summary.json $O/prof/conv_dw_c192_75_pmc.json
  bash tools/profile_cmd.sh conv_igemm32 c32 -- python tools/run_conv32.py 128 28 128 3 1 1 > $O/prof_c32.log 2>&1; cp gpurun_out/prof_c32/summary.json $O/prof/conv_igemm32_c128_28_pmc.json
  bash tools/profile_cmd.sh attention_kvcache_split kv -- python tools/kvcache_bench.py --splits -1 > $O/prof_kv.log 2>&1; cp gpurun_out/prof_kv/summary.json $O/prof/attention_kvcache_split_pmc.json
  echo "pmc done after $(( $(date +%s) - t0 )) s"
fi
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cp gpurun_out/bench_detail_n1.json $O/bench_detail.json
echo "bench exit $? after $(( $(date +%s) - t0 )) s"
timeout 300 python bench.py --warmup 5 --steps 20 > $O/bench_driverflags.json 2>> $O/bench.err; cp gpurun_out/bench_detail_n1.json $O/bench_driverflags_detail.json
for m in "resnet50 --tune" "resnet50 --idealised" "bert --tune" "bert --exporter hf4" "bert --decomposed" "bert --idealised" "llama" "llama --idealised"; do
  timeout 240 python tools/model_bench.py $m >> $O/models.json 2>> $O/models.err
done
timeout 300 python tools/membound_sweep.py --json $O/membound.json > $O/membound.txt 2>&1
timeout 300 python tools/conv_bench.py --variants=-1,2,7 > $O/conv_layers.txt 2>&1
timeout 200 python tools/conv_bench.py --variants=-1,2 --res --layers 3,7,13,19 > $O/conv_layers_residual.txt 2>&1
timeout 200 python tools/gemm_shapes.py --dtype bf16 --variants=-1,3,4,5,6 > $O/gemm_shapes_bf16.txt 2>&1
timeout 200 python tools/gemm_shapes.py --dtype f16 --variants=-1 > $O/gemm_shapes_f16.txt 2>&1
timeout 200 python tools/dwconv_bench.py > $O/dwconv_layers.txt 2>&1
timeout 200 python tools/conv32_bench.py > $O/conv32_layers.txt 2>&1
(for a in "--bh 32 --n 4096" "--bh 32 --n 32768" "--bh 256 --n 2048" "--bh 8 --n 8192 --d 256 --dtype bf16" "--bh 32 --n 4096 --dtype f32"; do timeout 100 python tools/kvcache_bench.py $a --splits 0,-1; done) > $O/kvcache.txt 2>&1
(timeout 100 python tools/conv_tap_timeline.py --wg 0; timeout 100 python tools/conv_tap_timeline.py --c 512 --h 7 --f 512 --wg 0) > $O/conv_tap_timeline.txt 2>&1
INFINI_ROCM_FUSION_LOG=1 timeout 200 python tools/model_bench.py resnet50 --iters 1 2> $O/resnet50_plan_log.txt > /dev/null
if [ -z "$NO_PROFILE" ]; then
  rm -rf gpurun_out/prof_models; bash tools/profile_models.sh > $O/prof_models.log 2>&1
  for f in $(find gpurun_out/prof_models -name "*kernel_stats.csv"); do cp $f $O/prof/; done
  cp gpurun_out/prof_models/bench_trace_summary.json $O/prof/ 2>/dev/null
fi
echo "total $(( $(date +%s) - t0 )) s"
tail -4 $O/pytest.log 2>/dev/null; tail -2 $O/smoke.log; cut -c1-600 $O/bench.json; echo; cut -c1-300 $O/bench_driverflags.json; echo; cut -c1-330 $O/models.json; tail -3 $O/conv_layers.txt
