#!/bin/bash
# Round 5, visit E: fp32 conv split-K — parity (forced and heuristic splits, the fp32 model tests), the per-layer forms, bench extras.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r5e
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout -k 10 400 python -m pytest tests/test_gpu_nn.py tests/test_gpu_models.py tests/test_gpu_frontend_exports.py -q -x -k "fp32 or f32 or reference_cpu_backend or export" > $O/pytest.log 2>&1
echo "pytest exit $? after $(( $(date +%s) - t0 )) s"; tail -4 $O/pytest.log
timeout -k 10 300 python tools/conv32_bench.py --forms > $O/conv32.txt 2>&1; cut -c1-60,150-330 $O/conv32.txt
timeout -k 10 240 python bench.py --no-graph --no-tp --no-cpu-baseline --steps 50 --warmup 50 > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; cut -c1-200 $O/bench.json; python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_detail_n1.json"))
print(json.dumps((d.get("extras") or {}).get("round5_kernels"), indent=0))
PY
echo "total $(( $(date +%s) - t0 )) s"
