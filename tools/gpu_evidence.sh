#!/bin/bash
# The evidence visit of a round: ONE box, everything the judge reads, written under gpurun_out/evidence/ with the names it keeps under
# profiles/ (tools/gpu_evidence.sh r06; then `cp gpurun_out/evidence/r06_* profiles/`). It has to be the LAST thing that touches csrc/:
# the counter files carry source stamps and tests/test_profile_stamps_cpu.py fails while one is stale.
# Every rocprofv3 step runs under `timeout -k`; PMC passes are separate runs with --kernel-trace only (never combined with other domains).
R=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/evidence
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
clean() { grep -v "amdgpu.ids\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl"; }
t0=$(date +%s)
step() { echo "== $1 after $(( $(date +%s) - t0 )) s" | tee -a $O/visit.log; }
# 1. parity suite + smoke
timeout -k 10 1500 python -m pytest tests -m gpu -q --durations=12 > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log; grep -E "passed|failed|error" $O/pytest.log | tail -3 > $O/${R}_gpu_suite_tail.txt; tail -1 $O/pytest.log >> $O/${R}_gpu_suite_tail.txt
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log
step "tests + smoke"
# 2. counter passes (stamped)
timeout -k 10 400 bash tools/profile_membound.sh > $O/prof_membound.log 2>&1 && cp gpurun_out/prof_membound/summary.json $O/${R}_membound_pmc.json
# (the headline GEMM on the kernel the heuristic launches for it — wave128, gemm128w.hip — and on the eight-wave kernel beside it)
timeout -k 10 400 bash tools/profile_gemm.sh 8 > $O/prof_gemm.log 2>&1 && cp gpurun_out/prof_gemm/summary.json $O/${R}_gemm_pmc.json
timeout -k 10 400 bash tools/profile_gemm.sh 4 > $O/prof_gemm4.log 2>&1 && cp gpurun_out/prof_gemm/summary.json $O/${R}_gemm256p_pmc.json
timeout -k 10 300 bash tools/profile_cmd.sh attention_kernel attn_bert -- python tools/attn_cmd.py > $O/prof_attn.log 2>&1 && cp gpurun_out/prof_attn_bert/summary.json $O/${R}_attention_bert_pmc.json
step "pmc"
# 3. the bench line (default flags, the driver's flags)
cd $REPO
timeout -k 10 700 python bench.py 2> $O/bench.err | clean | grep '^{' | tail -1 > $O/${R}_bench_line.json; cp gpurun_out/bench_detail_n1.json $O/${R}_bench_detail.json 2>/dev/null
timeout -k 10 400 python bench.py --warmup 5 --steps 20 2>> $O/bench.err | clean | grep '^{' | tail -1 > $O/${R}_bench_line_driverflags.json
step "bench"
# 4. graph-level lines and sweeps
for m in "resnet50 --tune" "bert --tune" "bert --decomposed" "llama"; do
  timeout -k 10 300 python tools/model_bench.py $m 2>> $O/models.err | clean | grep '^{' >> $O/${R}_model_lines.json
done
timeout -k 10 300 python tools/membound_sweep.py --json $O/${R}_membound_sweep.json 2>&1 | clean > $O/${R}_membound.txt
timeout -k 10 400 python tools/conv_bench.py --variants=-1,2,7 2>&1 | clean > $O/${R}_conv_layers.txt
timeout -k 10 200 python tools/conv_bench.py --variants=-1,2 --res --layers 3,7,13,19 2>&1 | clean > $O/${R}_conv_layers_residual.txt
timeout -k 10 300 python tools/gemm_shapes.py 2>&1 | clean > $O/${R}_gemm_shapes_bf16.txt
timeout -k 10 200 python tools/gemm_ktile_ledger.py 2>&1 | clean > $O/${R}_gemm_ktile_ledger.txt
timeout -k 10 500 python tools/gemm_wave128.py --ab 2>&1 | clean > $O/${R}_gemm_wave128_ab.txt
IROCM_W128_DBG=8 timeout -k 10 200 python tools/gemm_wave128.py --clock 2>&1 | clean > $O/${R}_gemm_wave128_clock.txt
step "models + sweeps"
# 5. kernel traces of the graphs and of the headline command
rm -rf gpurun_out/prof_models; timeout -k 10 700 bash tools/profile_models.sh > $O/prof_models.log 2>&1
for f in $(find gpurun_out/prof_models -name "*kernel_stats.csv"); do cp $f $O/${R}_$(basename $f); done
cp gpurun_out/prof_models/bench_trace_summary.json $O/${R}_bench_trace_summary.json 2>/dev/null
for m in "llama --iters 3" "bert --batch 32 --seq 512 --iters 3" "resnet50 --batch 128 --iters 3"; do
  echo "### model_bench.py $m: the kernels of ONE hipGraph replay" >> $O/${R}_graph_nodes.txt
  timeout -k 10 300 bash tools/graph_nodes.sh $m 2>&1 | clean | grep -v "^W2026\|^E2026" >> $O/${R}_graph_nodes.txt
done
step "traces"
echo "total $(( $(date +%s) - t0 )) s" | tee -a $O/visit.log
cat $O/${R}_gpu_suite_tail.txt; tail -2 $O/smoke.log; cut -c1-600 $O/${R}_bench_line_driverflags.json; echo; cut -c1-260 $O/${R}_model_lines.json
python3 -c "
import json;d=json.load(open('$O/${R}_membound_pmc.json'));print('membound stamp',d['stamp'],{k:v['traffic_over_algorithmic'] for k,v in list(d['rows'].items())[:6]})
g=json.load(open('$O/${R}_gemm_pmc.json'));print('gemm stamp',g.get('stamp'),'mfma busy',g.get('mfma_busy_frac'),'traffic',g.get('traffic_bytes_per_launch'))"
