#!/bin/bash
# round-2 final visit: everything gpu_round.sh does + the PMC passes of the persistent GEMM and the patch conv
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/round gpurun_out/prof_models
bash tools/gpu_round.sh > gpurun_out/round_stdout.log 2>&1
tail -14 gpurun_out/round_stdout.log | cut -c1-1500
bash tools/profile_gemm.sh 4 > gpurun_out/prof_gemm.log 2>&1; tail -3 gpurun_out/prof_gemm.log | cut -c1-1200
bash tools/profile_cmd.sh conv_patch patch -- python tools/conv_bench.py --layers 16 --variants=-1 --iters 5 > gpurun_out/prof_patch.log 2>&1; tail -12 gpurun_out/prof_patch.log
timeout 200 python tools/conv_bench.py --variants=-1 --iters 20 > gpurun_out/conv_layers.log 2>&1; tail -3 gpurun_out/conv_layers.log
bash tools/profile_cmd.sh attention_kernel attn_bert -- python tools/attn_cmd.py > gpurun_out/prof_attn.log 2>&1; tail -8 gpurun_out/prof_attn.log
python tools/gemm_shapes.py > gpurun_out/gemm_shapes.txt 2>&1; tail -9 gpurun_out/gemm_shapes.txt | cut -c1-260
python tools/gemm_timeline.py --wg 0,100 > gpurun_out/timeline_ffn1.txt 2>&1
python tools/gemm_timeline.py --m 16384 --n 768 --k 768 --tile 192 --wg 0,100 > gpurun_out/timeline_qkv.txt 2>&1
