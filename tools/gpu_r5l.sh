#!/bin/bash
# Round 5, visit L: tile-width rule of the pointwise conv mode — bench-size route tests, per-layer table, ResNet-50 graph line.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r5l
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout -k 10 400 python -m pytest tests/test_gpu_bench_routes.py tests/test_gpu_nn.py -q -x -k "route or pixel or pointwise or conv_mode or resnet or residual" > $O/pytest.log 2>&1
echo "pytest exit $? after $(( $(date +%s) - t0 )) s"; tail -3 $O/pytest.log | grep -v "version\|Hostname\|Librccl"
timeout -k 10 120 python tools/conv_bench.py --variants=-1 --res --layers 3,7,13,19 2>&1 | grep "^x\|total" | cut -c1-150
timeout -k 10 200 python tools/conv_bench.py --variants=-1 2>&1 | grep "^x\|total" | cut -c1-150
for i in 1 2; do timeout -k 10 200 python bench.py --no-tp --no-cpu-baseline --no-extras --steps 50 --warmup 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:v for k,v in d['config'].items() if 'resnet' in k}, d['value'])"; done
echo "total $(( $(date +%s) - t0 )) s"
