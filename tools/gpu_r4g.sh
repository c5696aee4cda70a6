#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r4g
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_nn.py tests/test_gpu_bench_routes.py tests/test_gpu_plugin.py tests/test_gpu_models.py -m gpu -q -x -k "stem or full_size" > $O/pytest_stem.log 2>&1
echo "pytest_stem exit $?" | tee -a $O/pytest_stem.log
timeout 120 python - > $O/stem_time.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from infinitensor_amd import RocmRuntime, ops
from infinitensor_amd.runtime import Event
from membound_sweep import timeit
rt = RocmRuntime(0); rt.use_torch_stream()
x = torch.rand(128, 3, 224, 224, device="cuda").half(); w = (torch.randn(64, 3, 7, 7, device="cuda") * 0.1).half(); b = torch.randn(64, device="cuda").half()
y = torch.empty(128, 64, 56, 56, device="cuda", dtype=torch.float16)
ops.set_conv_const_weights(rt, True)
t = timeit(rt, Event, lambda: ops.conv2d_pool(rt, x, w, b, 3, 3, 2, 2, 3, 2, 1, out=y))
print(f"stem+pool fused bs128: {t*1e6:.1f} us", flush=True)
PY
timeout 240 python tools/model_bench.py resnet50 >> $O/models.json 2>> $O/models.err
tail -5 $O/pytest_stem.log; cat $O/stem_time.txt; cut -c1-400 $O/models.json
