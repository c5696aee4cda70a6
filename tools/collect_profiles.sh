#!/bin/bash
# Copy the summaries of the last tools/gpu_round.sh visit (gpurun_out/round/) into profiles/ under this round's prefix.
R=${1:-r03}
cd $(dirname $0)/..
S=gpurun_out/round
clean() { grep -v "amdgpu.ids\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" "$1"; }
clean $S/bench.json | grep '^{' | tail -1 > profiles/${R}_bench_line.json
clean $S/bench_driverflags.json | grep '^{' | tail -1 > profiles/${R}_bench_line_driver_flags.json
grep '^{' $S/models.json > profiles/${R}_model_lines.json
for f in bench resnet50 bert bert_decomposed llama; do cp $S/prof/${f}_kernel_stats.csv profiles/${R}_${f}_kernel_stats.csv; done
cp $S/prof/bench_trace_summary.json profiles/${R}_bench_trace_summary.json
cp $S/prof/gemm256p_pmc.json profiles/${R}_gemm256p_pmc.json
cp $S/prof/gemm_fast32_pmc.json profiles/${R}_gemm_fast32_pmc.json
cp $S/prof/conv_pw_c512_f256_28_pmc.json profiles/${R}_conv_pw_c512_f256_28_pmc.json
cp $S/prof/conv_pw_c256_f1024_14_pmc.json profiles/${R}_conv_pw_c256_f1024_14_pmc.json
clean $S/conv_layers.txt > profiles/${R}_conv_layers.txt
clean $S/conv_layers_residual.txt > profiles/${R}_conv_layers_residual.txt
clean $S/conv_as_gemm.txt > profiles/${R}_conv_as_gemm.txt
clean $S/gemm_shapes_bf16.txt > profiles/${R}_gemm_shapes_bf16.txt
clean $S/gemm_timeline.txt > profiles/${R}_gemm_timeline.txt
clean $S/store_burst.txt > profiles/${R}_store_burst.txt
clean $S/store_burst2.txt > profiles/${R}_store_burst2.txt
clean $S/resnet50_fusion_log.txt > profiles/${R}_resnet50_plan_log.txt
grep '^{' $S/rocm_launch.log | tail -1 > profiles/${R}_rocm_launch_tp1_line.json
tail -4 $S/pytest.log | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" > profiles/${R}_pytest_gpu_tail.txt
ls -la profiles | grep ${R}_ | awk '{print $5, $9}'
