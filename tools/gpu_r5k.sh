#!/bin/bash
# Round 5, visit K: tile width of the pointwise conv mode on ResNet's residual-joined and plain pointwise layers (heuristic vs forced).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r5k
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
for nt in "" 4 3 2; do
  echo "== residual layers, IROCM_CONV_PW_NT=$nt"
  IROCM_CONV_PW_NT=$nt timeout -k 10 120 python tools/conv_bench.py --variants=-1 --res --layers 3,7,13,19 2>&1 | grep "^x" | cut -c1-150
done
for nt in "" 4 3 2; do
  echo "== plain pointwise layers, IROCM_CONV_PW_NT=$nt"
  IROCM_CONV_PW_NT=$nt timeout -k 10 120 python tools/conv_bench.py --variants=-1 --layers 3,5,7,9,11,13,14,15,17,19,20,21 2>&1 | grep "^x" | cut -c1-150
done
echo "total $(( $(date +%s) - t0 )) s"
