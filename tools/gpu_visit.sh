#!/bin/bash
# One parametrised GPU visit (replaces the per-round gpu_r4*.sh / gpu_r5*.sh one-offs): runs the named steps in order on ONE box and
# leaves everything under gpurun_out/<tag>/.   usage: tools/gpu_visit.sh TAG step [step ...]
# steps:  kvab            K-loop schedule variants (infinitensor_amd/lib/ab/kv*.so, tools/build_variant.py) against the shipped library:
#                         parity check + interleaved timing of the bf16 4096^3 headline + the fine K-tile ledger of each
#         tests:<expr>    python -m pytest tests -m gpu -x -q -k <expr>
#         testfile:<f>    python -m pytest tests/<f> -m gpu -x -q
#         bench           python bench.py (default flags) and with the driver's flags
#         py:<script+args>  python <script+args> (spaces as written, quote the step)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
TAG=$1; shift
O=gpurun_out/$TAG
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t00=$(date +%s)
for step in "$@"; do
  t0=$(date +%s)
  echo "==== step: $step" | tee -a $O/visit.log
  case "$step" in
    kvab)
      libs="base $(ls infinitensor_amd/lib/ab/ | grep '^kv' | sed 's/\.so$//')"
      for L in $libs; do
        if [ $L = base ]; then unset INFINI_ROCM_LIB; else export INFINI_ROCM_LIB=$REPO/infinitensor_amd/lib/ab/$L.so; fi
        echo "== $L check" | tee -a $O/kvab.txt
        timeout 300 python tools/gemm_ab.py --check 2>&1 | grep -v amdgpu.ids | tee -a $O/kvab.txt
      done
      for rep in 1 2; do
        for L in $libs; do
          if [ $L = base ]; then unset INFINI_ROCM_LIB; else export INFINI_ROCM_LIB=$REPO/infinitensor_amd/lib/ab/$L.so; fi
          echo "== $L rep $rep" | tee -a $O/kvab.txt
          timeout 300 python tools/gemm_ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/kvab.txt
        done
      done
      for L in $libs; do
        if [ $L = base ]; then unset INFINI_ROCM_LIB; else export INFINI_ROCM_LIB=$REPO/infinitensor_amd/lib/ab/$L.so; fi
        echo "== $L ledger (fine 4)" | tee -a $O/kv_ledger.txt
        timeout 300 python tools/gemm_ktile_ledger.py --fine 4 2>&1 | grep -v amdgpu.ids | tee -a $O/kv_ledger.txt
      done
      unset INFINI_ROCM_LIB
      ;;
    tests:*)
      timeout 3000 python -m pytest tests -m gpu -x -q -k "${step#tests:}" 2>&1 | tail -15 | tee -a $O/tests.txt
      ;;
    testfile:*)
      timeout 3000 python -m pytest tests/${step#testfile:} -m gpu -x -q 2>&1 | tail -15 | tee -a $O/tests.txt
      ;;
    bench)
      timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; cp gpurun_out/bench_detail_n1.json $O/bench_detail.json 2>/dev/null
      timeout -k 10 300 python bench.py --warmup 5 --steps 20 > $O/bench_driverflags.json 2>> $O/bench.err
      tail -c 1500 $O/bench_driverflags.json
      ;;
    py:*)
      timeout 1500 python ${step#py:} 2>&1 | grep -v amdgpu.ids | tee -a $O/py.txt
      ;;
    *) echo "unknown step $step";;
  esac
  echo "==== step $step: $(( $(date +%s) - t0 )) s" | tee -a $O/visit.log
done
echo "total $(( $(date +%s) - t00 )) s" | tee -a $O/visit.log
