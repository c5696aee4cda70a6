"""Softmax [196608, 512] (BASELINE config 4) time and HBM fraction per dtype.  python tools/softmax_probe.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from infinitensor_amd import RocmRuntime, ops
from infinitensor_amd.runtime import Event

rt = RocmRuntime(0)
for dt in (torch.float16, torch.bfloat16, torch.float32):
    x = torch.randn(196608, 512, device="cuda").to(dt)
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    for _ in range(5):
        ops.softmax(rt, x, 1, out=y)
    e0, e1 = Event(), Event()
    rt.record(e0)
    for _ in range(50):
        ops.softmax(rt, x, 1, out=y)
    rt.record(e1)
    us = rt.elapsed_ms(e0, e1) / 50 * 1e3
    nbytes = 2 * x.numel() * x.element_size()
    print(f"{str(dt):16s} {us:7.1f} us {nbytes / us / 1e3:7.1f} GB/s  {nbytes / us / 1e3 / 8000:.3f} of 8 TB/s")
