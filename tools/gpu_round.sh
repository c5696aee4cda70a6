#!/bin/bash
# One GPU visit of round 3: parity suite, smoke(), bench.py (default flags and the driver's), the model graphs in every lowering,
# kernel-trace profiles, PMC passes of the bf16 and fp32 headline GEMMs, the per-layer conv sweep.
# Everything lands under gpurun_out/round/ (copied to profiles/r03_* by hand afterwards).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/round
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout ${PYTEST_LIMIT:-900} python -m pytest tests -m gpu -q --durations=12 > $O/pytest.log 2>&1
echo "pytest exit $? after $(( $(date +%s) - t0 )) s" | tee -a $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke exit $?" | tee -a $O/smoke.log
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench exit $? after $(( $(date +%s) - t0 )) s"
timeout 500 python bench.py --warmup 5 --steps 20 --no-graph --no-tp --no-cpu-baseline --no-extras > $O/bench_driverflags.json 2>> $O/bench.err
for m in "resnet50 --tune" "resnet50 --idealised" "bert --tune" "bert --idealised" "bert --decomposed" "bert --merged-kt" "llama" "llama --idealised"; do
  timeout 240 python tools/model_bench.py $m >> $O/models.json 2>> $O/models.err
done
(cd $O && timeout 300 python $REPO/tools/rocm_launch.py --nproc_per_node 1 > rocm_launch.log 2>&1; echo "rocm_launch exit $?" >> rocm_launch.log)
INFINI_ROCM_FUSION_LOG=1 timeout 200 python tools/model_bench.py resnet50 --iters 1 2> $O/resnet50_fusion_log.txt > /dev/null
timeout 300 python tools/conv_bench.py --variants=-1,2,5 > $O/conv_layers.txt 2>&1
timeout 200 python tools/conv_bench.py --variants=-1,2 --res --layers 3,7,13,19 > $O/conv_layers_residual.txt 2>&1
for s in "2048 2048" "16384 3072"; do set -- $s; echo "== m $1 n $2 k 768"; timeout 100 python tools/gemm_timeline.py --m $1 --n $2 --k 768 --wg 0 2>&1 | grep -v amdgpu; done > $O/gemm_timeline.txt
(hipcc --offload-arch=gfx950 -O2 -w tools/probes/store_burst2.hip -o /tmp/sb2 && timeout 100 /tmp/sb2) > $O/store_burst2.txt 2>&1
(hipcc --offload-arch=gfx950 -O2 -w tools/probes/store_burst.hip -o /tmp/sb1 && timeout 100 /tmp/sb1) > $O/store_burst.txt 2>&1
timeout 200 python tools/probes/conv_as_gemm.py > $O/conv_as_gemm.txt 2>&1
timeout 200 python tools/gemm_shapes.py --dtype bf16 > $O/gemm_shapes_bf16.txt 2>&1
if [ -z "$NO_PROFILE" ]; then
  bash tools/profile_models.sh > $O/prof.log 2>&1
  mkdir -p $O/prof
  for f in $(find gpurun_out/prof_models -name "*kernel_stats.csv"); do cp $f $O/prof/; done
  cp gpurun_out/prof_models/bench_trace_summary.json $O/prof/ 2>/dev/null
  bash tools/profile_gemm.sh 4 > $O/prof_gemm.log 2>&1; cp gpurun_out/prof_gemm/summary.json $O/prof/gemm256p_pmc.json
  bash tools/profile_cmd.sh gemm_fast32 gemm32 -- python tools/run_gemm32.py 4096 5 > $O/prof_gemm32.log 2>&1; cp gpurun_out/prof_gemm32/summary.json $O/prof/gemm_fast32_pmc.json
  # the conv mode of the persistent GEMM on two pointwise layers: C512 -> F256 @28x28 (index 11) and C256 -> F1024 @14x14 (index 13)
  bash tools/profile_cmd.sh gemm256p_kernel conv_pw_c512_28 -- python tools/conv_bench.py --variants=-1 --layers 11 --iters 5 > $O/prof_conv_pw1.log 2>&1; cp gpurun_out/prof_conv_pw_c512_28/summary.json $O/prof/conv_pw_c512_f256_28_pmc.json
  bash tools/profile_cmd.sh gemm256p_kernel conv_pw_c256_14 -- python tools/conv_bench.py --variants=-1 --layers 13 --iters 5 > $O/prof_conv_pw2.log 2>&1; cp gpurun_out/prof_conv_pw_c256_14/summary.json $O/prof/conv_pw_c256_f1024_14_pmc.json
fi
echo "total $(( $(date +%s) - t0 )) s"
tail -3 $O/pytest.log; tail -2 $O/smoke.log; cut -c1-600 $O/bench.json; cut -c1-400 $O/bench_driverflags.json; cat $O/models.json | cut -c1-420
