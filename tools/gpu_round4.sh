#!/bin/bash
# One GPU visit of round 4: parity suite, smoke(), bench.py (default flags and the driver's), model graphs, kernel traces, PMC passes
# (headline GEMM, memory-bound sweep, the fused stem), per-layer conv sweep, GEMM shapes. Everything lands under gpurun_out/round4/
# (copied to profiles/r04_* afterwards). NO_PYTEST=1 / NO_PROFILE=1 skip those parts.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/round4
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
  cp $O/prof/membound_pmc.json profiles/r04_membound_pmc.json; cp $O/prof/gemm256p_pmc.json profiles/r04_gemm256p_pmc.json
  bash tools/profile_cmd.sh conv_stem_pool stem -- python tools/run_stem.py > $O/prof_stem.log 2>&1; cp gpurun_out/prof_stem/summary.json $O/prof/conv_stem_pool_pmc.json
  echo "pmc done after $(( $(date +%s) - t0 )) s"
fi
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench exit $? after $(( $(date +%s) - t0 )) s"
timeout 300 python bench.py --warmup 5 --steps 20 --no-graph --no-tp --no-cpu-baseline --no-extras > $O/bench_driverflags.json 2>> $O/bench.err
for m in "resnet50 --tune" "resnet50 --idealised" "bert --tune" "bert --exporter hf4" "bert --decomposed" "bert --idealised" "llama" "llama --idealised"; do
  timeout 240 python tools/model_bench.py $m >> $O/models.json 2>> $O/models.err
done
timeout 300 python tools/membound_sweep.py --json $O/membound.json > $O/membound.txt 2>&1
timeout 300 python tools/conv_bench.py --variants=-1,2 > $O/conv_layers.txt 2>&1
timeout 200 python tools/conv_bench.py --variants=-1,2 --res --layers 3,7,13,19 > $O/conv_layers_residual.txt 2>&1
timeout 200 python tools/gemm_shapes.py --dtype bf16 --variants=-1,3,4,5,6 > $O/gemm_shapes_bf16.txt 2>&1
timeout 200 python tools/gemm_shapes.py --dtype f16 --variants=-1 > $O/gemm_shapes_f16.txt 2>&1
INFINI_ROCM_FUSION_LOG=1 timeout 200 python tools/model_bench.py resnet50 --iters 1 2> $O/resnet50_plan_log.txt > /dev/null
if [ -z "$NO_PROFILE" ]; then
  rm -rf gpurun_out/prof_models; bash tools/profile_models.sh > $O/prof_models.log 2>&1
  for f in $(find gpurun_out/prof_models -name "*kernel_stats.csv"); do cp $f $O/prof/; done
  cp gpurun_out/prof_models/bench_trace_summary.json $O/prof/ 2>/dev/null
fi
echo "total $(( $(date +%s) - t0 )) s"
tail -4 $O/pytest.log 2>/dev/null; tail -2 $O/smoke.log; cut -c1-500 $O/bench.json; echo; cut -c1-300 $O/bench_driverflags.json; echo; cut -c1-330 $O/models.json; tail -3 $O/conv_layers.txt; cat $O/gemm_shapes_bf16.txt | cut -c1-200
