#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r4j
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_multi.py -m gpu -q -x -k "bench_tp_block" > $O/pytest.log 2>&1
echo "pytest exit $?" | tee -a $O/pytest.log
tail -40 $O/pytest.log
IROCM_BENCH_ONE_DEVICE=1 INFINI_ROCM_COMM=direct timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-extras > $O/bench2.json 2> $O/bench2.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4j/bench2.json").read().strip().splitlines()[-1])
    print(json.dumps(d.get("tp_block"))[:1500])
except Exception as e:
    print("no line", e); print(open("gpurun_out/r4j/bench2.err").read()[-2000:])
PY
