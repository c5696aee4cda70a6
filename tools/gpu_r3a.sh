#!/bin/bash
# round 3, visit a: the new launch planner on hardware — parity suite, fuzzers, graphs in the front-end's lowering
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r3a
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q --durations=8 -x > $O/pytest.log 2>&1
echo "pytest exit $? after $(( $(date +%s) - t0 )) s" | tee -a $O/pytest.log
tail -25 $O/pytest.log
for m in "resnet50" "resnet50 --idealised" "bert" "bert --idealised" "bert --decomposed" "bert --merged-kt" "llama" "llama --idealised"; do
  timeout 200 python tools/model_bench.py $m >> $O/models.json 2>> $O/models.err
done
cat $O/models.json
FUZZ_SEED=77 timeout 300 python tools/onnx_form_fuzz.py 60 > $O/fuzz_onnx.log 2>&1; tail -3 $O/fuzz_onnx.log
FUZZ_SEED=78 INFINI_ROCM_FUSE_GELU=0 timeout 300 python tools/fusion_fuzz.py 60 > $O/fuzz_fusion.log 2>&1; tail -2 $O/fuzz_fusion.log
FUZZ_SEED=79 timeout 300 python tools/conv_fusion_fuzz.py 40 > $O/fuzz_conv.log 2>&1; tail -2 $O/fuzz_conv.log
timeout 400 python bench.py --warmup 5 --steps 20 > $O/bench_driverflags.json 2> $O/bench.err
echo "bench exit $?"; cut -c1-900 $O/bench_driverflags.json
echo "total $(( $(date +%s) - t0 )) s"
