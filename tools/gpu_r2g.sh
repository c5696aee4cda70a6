#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2g
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_nn.py -m gpu -q --maxfail=12 -p no:cacheprovider -k "conv" > gpurun_out/r2g/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2g/pytest.log
tail -25 gpurun_out/r2g/pytest.log
timeout 300 python tools/conv_bench.py --variants=-1,4 --iters 20 --layers 2,6,10,12,16,18,22 > gpurun_out/r2g/conv_layers.log 2>&1; cat gpurun_out/r2g/conv_layers.log
