"""Fixed cost of the 256^2 GEMM at the headline grid: time vs K (slope = per-K-tile time, intercept = launch +
prologue + epilogue)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from infinitensor_amd import RocmRuntime, ops
from infinitensor_amd.runtime import Event

rt = RocmRuntime(0)
ops.set_matmul_variant(rt, 2)
for k in (64, 128, 256, 512, 1024, 2048, 4096):
    a = torch.randn(4096, k, device="cuda").to(torch.bfloat16)
    b = torch.randn(k, 4096, device="cuda").to(torch.bfloat16)
    c = torch.empty(4096, 4096, device="cuda", dtype=torch.bfloat16)
    torch.cuda.synchronize()
    for _ in range(5):
        ops.matmul(rt, a, b, out=c)
    e0, e1 = Event(), Event()
    rt.record(e0)
    for _ in range(50):
        ops.matmul(rt, a, b, out=c)
    rt.record(e1)
    us = rt.elapsed_ms(e0, e1) / 50 * 1e3
    print(f"k={k:5d} tiles={k // 64:3d}  {us:8.2f} us  {us / (k // 64):6.3f} us/tile")
