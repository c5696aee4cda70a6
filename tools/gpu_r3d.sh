#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r3d
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_plugin.py tests/test_gpu_models.py tests/test_gpu_fuzz.py -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest exit $?"; tail -5 $O/pytest.log
FUZZ_SEED=91 timeout 300 python tools/conv_fusion_fuzz.py 80 > $O/fuzz_conv.log 2>&1; tail -2 $O/fuzz_conv.log
FUZZ_SEED=92 timeout 300 python tools/onnx_form_fuzz.py 80 > $O/fuzz_onnx.log 2>&1; tail -2 $O/fuzz_onnx.log
FUZZ_SEED=93 INFINI_ROCM_FUSE_GELU=0 timeout 300 python tools/fusion_fuzz.py 80 > $O/fuzz_fusion.log 2>&1; tail -1 $O/fuzz_fusion.log
timeout 200 python tools/model_bench.py resnet50 > $O/rn.json 2>$O/rn.err; cat $O/rn.json
echo "total $(( $(date +%s) - t0 )) s"
