#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2e
export TMPDIR=/tmp
for cfg in "4 0" "8 0" "4 2" "8 2" "2 0" "16 0"; do set -- $cfg
  echo "== IROCM_NORM_BLOCKS_PER_CU=$1 IROCM_NORM_RPW=$2"
  LN_SHAPES=16384x768,262144x768 IROCM_NORM_BLOCKS_PER_CU=$1 IROCM_NORM_RPW=$2 timeout 120 python tools/ln_probe.py 2>&1 | grep "^LN"
done > gpurun_out/r2e/ln_sweep.log 2>&1
cat gpurun_out/r2e/ln_sweep.log
timeout 600 bash tools/profile_rowops.sh > gpurun_out/r2e/rowops_pmc.log 2>&1; tail -20 gpurun_out/r2e/rowops_pmc.log
