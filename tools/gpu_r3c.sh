#!/bin/bash
# round 3, visit c: fp32 tile kernel, overlapped collectives (world 1), bench extras
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r3c
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_matmul.py tests/test_gpu_multi.py tests/test_gpu_rowops.py -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest exit $?"; tail -15 $O/pytest.log
timeout 400 python bench.py --no-graph > $O/bench.json 2> $O/bench.err; echo "bench exit $?"
python - <<PY
import json
d = json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
print(d["value"], d["roofline"]["frac"], d["roofline"]["cold20"])
print(json.dumps(d["tp_block"])[:1500])
print({k: v for k, v in d.get("extras", {}).items() if "f32" in k})
PY
echo "total $(( $(date +%s) - t0 )) s"
