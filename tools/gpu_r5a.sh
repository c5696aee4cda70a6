#!/bin/bash
# Round 5, visit A: the tap mode (3 x 3 layers as one GEMM on the persistent kernels) — parity tests, then the per-layer sweep of
# ResNet-50's 3 x 3 layers (heuristic route vs variant 7 at every tile width).  FULL=1 also runs the whole GPU suite.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r5a
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
[ -n "$NO_TAP_TESTS" ] || timeout 900 python -m pytest tests/test_gpu_nn.py -q -x -k "tap_gemm" > $O/pytest_tap.log 2>&1
echo "tap tests exit $? after $(( $(date +%s) - t0 )) s" | tee -a $O/pytest_tap.log
tail -25 $O/pytest_tap.log
timeout 300 python tools/conv_bench.py --variants=-1,7 --tap-nt 0,4 --layers 12,16,18,22 > $O/conv3x3.txt 2>&1
echo "conv bench exit $? after $(( $(date +%s) - t0 )) s"
cat $O/conv3x3.txt | cut -c1-400
if [ -n "$FULL" ]; then
  timeout 1200 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1
  echo "pytest exit $? after $(( $(date +%s) - t0 )) s" | tee -a $O/pytest.log
  tail -15 $O/pytest.log
fi
echo "total $(( $(date +%s) - t0 )) s"
