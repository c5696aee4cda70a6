#!/bin/bash
# Copy the summaries of the last tools/gpu_round5.sh visit (gpurun_out/round5/) into profiles/ under this round's prefix.
R=${1:-r05}
cd $(dirname $0)/..
S=gpurun_out/round5
clean() { grep -v "amdgpu.ids\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" "$1"; }
clean $S/bench.json | grep '^{' | tail -1 > profiles/${R}_bench_line.json
clean $S/bench_driverflags.json | grep '^{' | tail -1 > profiles/${R}_bench_line_driverflags.json
cp $S/bench_detail.json profiles/${R}_bench_detail.json
grep '^{' $S/models.json > profiles/${R}_model_lines.json
# (a lean visit — tools/gpu_round5_final.sh — re-takes only what changed: files it did not produce keep their earlier version)
for f in bench resnet50 bert bert_decomposed llama; do [ -f $S/prof/${f}_kernel_stats.csv ] && cp $S/prof/${f}_kernel_stats.csv profiles/${R}_${f}_kernel_stats.csv; done
[ -f $S/prof/bench_trace_summary.json ] && cp $S/prof/bench_trace_summary.json profiles/${R}_bench_trace_summary.json
for f in gemm256p_pmc membound_pmc conv_stem_pool_pmc conv_tap_splitk_c512_7_pmc conv_dw_c192_75_pmc conv_igemm32_c128_28_pmc attention_kvcache_split_pmc; do
  [ -f $S/prof/$f.json ] && cp $S/prof/$f.json profiles/${R}_$f.json
done
for f in conv_layers conv_layers_residual gemm_shapes_bf16 gemm_shapes_f16 dwconv_layers conv32_layers kvcache conv_tap_timeline membound stem; do
  [ -f $S/$f.txt ] && clean $S/$f.txt > profiles/${R}_$f.txt
done
cp $S/membound.json profiles/${R}_membound_sweep.json
clean $S/resnet50_plan_log.txt > profiles/${R}_resnet50_plan_log.txt
clean $S/pytest.log > profiles/${R}_pytest_gpu.log
tail -4 $S/pytest.log | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" > profiles/${R}_gpu_suite_tail.txt
ls -la profiles | grep ${R}_ | awk '{print $5, $9}'
