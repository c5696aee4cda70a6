#!/bin/bash
# round 4, visit B: the hand-written transport at world 2 / 4 / 8 on one device, the memory-bound sweep, Gelu epilogue timing,
# BERT / Llama graph times, then the GPU suite.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r4b
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_multi.py -m gpu -q -x --durations=8 > $O/pytest_multi.log 2>&1
echo "pytest_multi exit $? after $(( $(date +%s) - t0 )) s" | tee -a $O/pytest_multi.log
timeout 600 python tools/membound_sweep.py --json $O/membound.json > $O/membound.txt 2>&1
echo "membound exit $? after $(( $(date +%s) - t0 )) s"
timeout 120 python - > $O/gelu_ffn1.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, ".")
from infinitensor_amd import RocmRuntime, ops
from infinitensor_amd.runtime import Event
sys.path.insert(0, "tools")
from membound_sweep import timeit
rt = RocmRuntime(0); rt.use_torch_stream()
a = torch.randn(16384, 768, device="cuda").half(); w = (torch.randn(768, 3072, device="cuda") * 0.05).half(); b = torch.randn(3072, device="cuda").half()
c = torch.empty(16384, 3072, device="cuda", dtype=torch.float16)
for act in (0, 5, 4, 1):
    t = timeit(rt, Event, lambda: ops.matmul(rt, a, w, b, act=act, out=c))
    print(f"ffn1 16384x3072x768 f16 act={act}: {t * 1e6:.1f} us ({ops.matmul_last_variant(rt)})", flush=True)
PY
for m in "bert" "bert --decomposed" "llama" "resnet50"; do
  timeout 240 python tools/model_bench.py $m >> $O/models.json 2>> $O/models.err
done
echo "models done after $(( $(date +%s) - t0 )) s"
timeout 900 python -m pytest tests -m gpu -q --durations=8 --deselect tests/test_gpu_multi.py > $O/pytest_rest.log 2>&1
echo "pytest_rest exit $? after $(( $(date +%s) - t0 )) s" | tee -a $O/pytest_rest.log
tail -30 $O/pytest_multi.log; cat $O/membound.txt; cat $O/gelu_ffn1.txt; cut -c1-500 $O/models.json; tail -8 $O/pytest_rest.log
