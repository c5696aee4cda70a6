"""LayerNorm / Softmax timing probe (HIP events on the runtime stream)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from infinitensor_amd import RocmRuntime, ops
from infinitensor_amd.runtime import Event

rt = RocmRuntime(0)
import os

SHAPES = ((16384, 768), (262144, 768), (16384, 4096))
if os.environ.get("LN_SHAPES"):  # e.g. LN_SHAPES=16384x768
    SHAPES = tuple(tuple(int(v) for v in s.split("x")) for s in os.environ["LN_SHAPES"].split(","))
for rows, n in SHAPES:
    for dt in (torch.float16, torch.float32):
        x = torch.randn(rows, n, device="cuda").to(dt)
        g = torch.randn(n, device="cuda").to(dt)
        b = torch.randn(n, device="cuda").to(dt)
        y = torch.empty_like(x)
        torch.cuda.synchronize()
        for _ in range(5):
            ops.layer_norm(rt, x, g, b, 1e-5, -1, out=y)
        e0, e1 = Event(), Event()
        rt.record(e0)
        for _ in range(50):
            ops.layer_norm(rt, x, g, b, 1e-5, -1, out=y)
        rt.record(e1)
        rt.sync()
        us = rt.elapsed_ms(e0, e1) / 50 * 1e3
        print(f"LN {rows}x{n} {str(dt)[6:]:8s} {us:8.2f} us  {2 * x.numel() * x.element_size() / us / 1e3:8.1f} GB/s", flush=True)
