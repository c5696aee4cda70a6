"""What the GEMM machinery would do on ResNet-50's pointwise layers if they were plain GEMMs W[F x C] . X[C x (N HW)]:
an upper bound for a conv mode of each kernel family (fast128 / tile256 / persistent 256-192-128), next to today's conv path.
python tools/probes/conv_as_gemm.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from infinitensor_amd import RocmRuntime, ops
from infinitensor_amd.runtime import Event

LAYERS = [(64, 56, 256), (128, 28, 512), (256, 14, 1024), (512, 7, 2048), (1024, 14, 256), (512, 28, 256), (1024, 14, 512),
          (2048, 7, 512), (256, 56, 128), (512, 28, 128), (256, 56, 64)]
rt = RocmRuntime(0)
names = ops.matmul_variants()
dt = torch.float16
NB = 128


def timeit(fn, iters=30):
    for _ in range(10):
        fn()
    e0, e1 = Event(), Event()
    rt.record(e0)
    for _ in range(iters):
        fn()
    rt.record(e1)
    rt.sync()
    return rt.elapsed_ms(e0, e1) / iters * 1e3


x0 = torch.randn(4096, 4096, device="cuda").to(dt)
for _ in range(300):  # clocks up
    ops.matmul(rt, x0, x0)
for c, hw, f in LAYERS:
    n = NB * hw * hw
    n8 = (n + 7) // 8 * 8
    w = (torch.randn(f, c, device="cuda") / c ** 0.5).to(dt)
    x = torch.randn(c, n8, device="cuda").to(dt)
    y = torch.empty(f, n8, device="cuda", dtype=dt)
    nbytes = 2.0 * (c * n + f * c + f * n)
    flop = 2.0 * f * c * n
    floor = max(nbytes / 6.3e12, flop / 2.15e15) * 1e6
    line = f"C{c:<4d} {hw:>2d}x{hw:<2d} F{f:<4d} floor {floor:5.1f} us |"
    for v in (-1, 1, 2, 4, 5, 6):
        ops.set_matmul_variant(rt, v)
        us = timeit(lambda: ops.matmul(rt, w, x, out=y))
        line += f" {'heur' if v < 0 else names[v][:10]:>10s} {us:6.1f} x{us / floor:3.1f} |"
    ops.set_matmul_variant(rt, -1)
    xc = torch.randn(NB, c, hw, hw, device="cuda").to(dt)
    wc = w.view(f, c, 1, 1).contiguous()
    b = torch.randn(f, device="cuda").to(dt)
    yc = torch.empty(NB, f, hw, hw, device="cuda", dtype=dt)
    us = timeit(lambda: ops.conv2d(rt, xc, wc, 0, 0, 1, 1, bias=b, act=1, out=yc))
    line += f" conv today {us:6.1f} x{us / floor:3.1f}"
    print(line, flush=True)
