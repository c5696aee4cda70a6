#!/bin/bash
# round 4, visit D: collective tests (direct transport, worlds 2 / 4 / 8), the whole GPU suite, the memory-bound sweep after the
# kernel fixes and its PMC traffic, BERT in the real export's order.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r4d
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $O/pytest.log 2>&1
echo "pytest exit $? after $(( $(date +%s) - t0 )) s" | tee -a $O/pytest.log
timeout 600 python tools/membound_sweep.py --json $O/membound.json > $O/membound.txt 2>&1
echo "membound exit $? after $(( $(date +%s) - t0 )) s"
timeout 900 bash tools/profile_membound.sh > $O/prof_membound.log 2>&1
cp gpurun_out/prof_membound/summary.json $O/membound_pmc.json 2>/dev/null
echo "prof_membound done after $(( $(date +%s) - t0 )) s"
for m in "bert" "bert --exporter hf4" "bert --decomposed" "llama" "resnet50"; do
  timeout 240 python tools/model_bench.py $m >> $O/models.json 2>> $O/models.err
done
echo "models done after $(( $(date +%s) - t0 )) s"
tail -25 $O/pytest.log; cat $O/membound.txt; tail -5 $O/prof_membound.log; cut -c1-420 $O/models.json
