#!/bin/bash
# Round 5, visit B: the whole GPU suite after the shaped-glue move, then the decode-attention, fp32-conv and depthwise sweeps.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=gpurun_out/r5b
rm -rf $O; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest.log 2>&1
echo "pytest exit $? after $(( $(date +%s) - t0 )) s" | tee -a $O/pytest.log
tail -12 $O/pytest.log
for sh in "--bh 32 --n 4096" "--bh 32 --n 32768" "--bh 256 --n 2048" "--bh 8 --n 8192 --d 256 --dtype bf16" "--bh 32 --n 4096 --dtype f32"; do
  timeout 100 python tools/kvcache_bench.py $sh --splits=-1 >> $O/kvcache.txt 2>&1
  IROCM_KVCACHE_TWO_LAUNCH=1 timeout 100 python tools/kvcache_bench.py $sh --splits=-1 >> $O/kvcache_two_launch.txt 2>&1
done
cat $O/kvcache.txt; echo "-- two launches:"; cat $O/kvcache_two_launch.txt
timeout 300 python tools/conv32_bench.py > $O/conv32.txt 2>&1; head -3 $O/conv32.txt; tail -2 $O/conv32.txt
timeout 300 python tools/dwconv_bench.py --th-mults 1,2,3,4 > $O/dw.txt 2>&1; cat $O/dw.txt | cut -c1-260
echo "total $(( $(date +%s) - t0 )) s"
