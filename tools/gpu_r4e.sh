#!/bin/bash
# round 4, visit E: the fused stem (tests first), then ResNet-50 graph time, the remaining suite.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r4e
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_nn.py tests/test_gpu_bench_routes.py tests/test_gpu_plugin.py tests/test_gpu_matmul.py tests/test_gpu_models.py -m gpu -q -x -k "stem or reduce or compute_type or full_size or frontend" --durations=5 > $O/pytest_stem.log 2>&1
echo "pytest_stem exit $? after $(( $(date +%s) - t0 )) s" | tee -a $O/pytest_stem.log
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
c = torch.empty(128, 64, 112, 112, device="cuda", dtype=torch.float16)
t1 = timeit(rt, Event, lambda: ops.conv2d(rt, x, w, 3, 3, 2, 2, bias=b, act=1, out=c))
t2 = timeit(rt, Event, lambda: ops.max_pool(rt, c, 3, 3, 1, 1, 1, 1, 2, 2, 0, out=y))
print(f"unfused: conv {t1*1e6:.1f} us + pool {t2*1e6:.1f} us", flush=True)
PY
for m in "resnet50" "bert" "llama"; do
  timeout 240 python tools/model_bench.py $m >> $O/models.json 2>> $O/models.err
done
echo "models done after $(( $(date +%s) - t0 )) s"
timeout 1500 python -m pytest tests -m gpu -q --durations=6 --deselect tests/test_gpu_multi.py > $O/pytest_rest.log 2>&1
echo "pytest_rest exit $? after $(( $(date +%s) - t0 )) s" | tee -a $O/pytest_rest.log
tail -30 $O/pytest_stem.log; cat $O/stem_time.txt; cut -c1-420 $O/models.json; tail -12 $O/pytest_rest.log
