#!/bin/bash
# Round 5, visit D: stem tile order A/B, ResNet line A/B + kernel trace, fp32 conv tests + sweep with the new routing.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r5d
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
for i in 1 2; do
  python tools/run_stem.py 2>&1 | tail -1
  IROCM_STEM_LINEAR=1 python tools/run_stem.py 2>&1 | tail -1 | sed 's/^/linear order: /'
done
timeout 300 python tools/model_bench.py resnet50 2>&1 | tail -1 | cut -c1-60,250-420
IROCM_STEM_LINEAR=1 timeout 300 python tools/model_bench.py resnet50 2>&1 | tail -1 | cut -c1-60,250-420
timeout 600 python -m pytest tests/test_gpu_nn.py -q -x -k "fp32" > $O/pytest_fp32.log 2>&1; tail -3 $O/pytest_fp32.log
timeout 400 python tools/conv32_bench.py --forms > $O/conv32.txt 2>&1; cat $O/conv32.txt | cut -c1-250
cd /tmp && export TMPDIR=/tmp
# (as first run this line had no --output-format csv and a plain `timeout`: rocprofv3's post-processing of the default database did not
# end and ignored SIGTERM — the visit ran into gpurun's own limit. Every profiler step of the later scripts is `timeout -k` + csv.)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof_resnet -o resnet -- python $REPO/tools/model_bench.py resnet50 --iters 5 > $REPO/$O/prof_resnet.log 2>&1
cd $REPO
f=$(find $O/prof_resnet -name "*kernel_stats.csv" | head -1); echo $f; head -25 $f | cut -c1-200
echo "total $(( $(date +%s) - t0 )) s"
