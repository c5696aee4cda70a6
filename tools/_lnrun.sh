timeout 900 python -m pytest tests/test_gpu_rowops.py tests/test_gpu_plugin.py -q 2>&1 | tail -5
python tools/ln_probe.py 2>&1 | grep LN
python tools/model_bench.py bert 2>&1 | tail -1
