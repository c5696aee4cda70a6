"""The HBM-roofline rows of SURVEY 8d (C4 shapes), a few launches each — the command tools/profile_rowops.sh profiles."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from infinitensor_amd import RocmRuntime, ops

rt = RocmRuntime(0)
for dt in (torch.float16, torch.float32):
    x = torch.randn(196608, 512, device="cuda").to(dt)
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    for _ in range(6):
        ops.softmax(rt, x, 1, out=y)
    rt.sync()
    del x, y
    for rows in (16384, 262144):
        x = torch.randn(rows, 768, device="cuda").to(dt)
        g = torch.randn(768, device="cuda").to(dt)
        b = torch.randn(768, device="cuda").to(dt)
        y = torch.empty_like(x)
        torch.cuda.synchronize()
        for _ in range(6):
            ops.layer_norm(rt, x, g, b, 1e-5, -1, out=y)
        rt.sync()
        del x, y
