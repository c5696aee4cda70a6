#!/bin/bash
# A/B in ONE visit (boxes differ by several per cent): infinitensor_amd/lib/ab/base.so against the current library, interleaved.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/ab
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do
  for which in base new; do
    if [ $which = base ]; then export INFINI_ROCM_LIB=$REPO/infinitensor_amd/lib/ab/base.so; else unset INFINI_ROCM_LIB; fi
    for cmd in "$@"; do
      echo "== $which rep $rep: $cmd" | tee -a $O/ab.txt
      timeout 300 python $cmd 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
    done
  done
done
