#!/bin/bash
# One GPU visit: parity suite, smoke(), bench.py, BERT through the executor (with h.tune()), kernel-trace profiles.
# Everything lands under gpurun_out/round/ (copied to profiles/ by hand afterwards).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/round
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout ${PYTEST_LIMIT:-900} python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest.log 2>&1
echo "pytest exit $? after $(( $(date +%s) - t0 )) s" | tee -a $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke exit $?" | tee -a $O/smoke.log
timeout 420 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench exit $? after $(( $(date +%s) - t0 )) s"
timeout 240 python tools/model_bench.py bert --tune > $O/bert.json 2> $O/bert.err
timeout 240 python tools/model_bench.py llama > $O/llama.json 2> $O/llama.err
(cd $O && timeout 300 python $REPO/tools/rocm_launch.py --nproc_per_node 1 > rocm_launch.log 2>&1; echo "rocm_launch exit $?" >> rocm_launch.log)
if [ -z "$NO_PROFILE" ]; then
  bash tools/profile_models.sh > $O/prof.log 2>&1
  mkdir -p $O/prof
  for f in $(find gpurun_out/prof_models -name "*kernel_stats.csv"); do cp $f $O/prof/; done
fi
echo "total $(( $(date +%s) - t0 )) s"
tail -3 $O/pytest.log; tail -2 $O/smoke.log; cat $O/bench.json | cut -c1-1500; grep model $O/bert.json; grep model $O/llama.json; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/rocm_launch.log | tail -6
