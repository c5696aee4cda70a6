"""GEMM time per kernel variant over the shapes of BASELINE configs 4 and 5 (BERT-base bs32 seq512 projections / FFN,
Llama-7B block at 2048 tokens) and the headline 4096^3.  python tools/gemm_shapes.py [--dtype f16] [--variants -1,1,2,3,4,5,6]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from infinitensor_amd import RocmRuntime, ops
from infinitensor_amd.runtime import Event

SHAPES = [  # (name, m, n, k, bias)
    ("bert qkv/out", 16384, 768, 768, True), ("bert ffn1", 16384, 3072, 768, True), ("bert ffn2", 16384, 768, 3072, True),
    ("llama qkv/o", 2048, 4096, 4096, False), ("llama gate/up", 2048, 11008, 4096, False), ("llama down", 2048, 4096, 11008, False),
    ("headline", 4096, 4096, 4096, False), ("8192^3", 8192, 8192, 8192, False),
]

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f16")
ap.add_argument("--variants", default="-1,1,2,3,4,5,6")
ap.add_argument("--iters", type=int, default=50)
args = ap.parse_args()
dt = {"f16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
rt = RocmRuntime(0)
names = ops.matmul_variants()
for name, m, n, k, has_bias in SHAPES:
    a = torch.randn(m, k, device="cuda").to(dt)
    b = (torch.randn(k, n, device="cuda") * 0.05).to(dt)
    bias = torch.randn(n, device="cuda").to(dt) if has_bias else None
    c = torch.empty(m, n, device="cuda", dtype=dt)
    torch.cuda.synchronize()
    # the operands were just created: the first variant of a row must not pay the clock ramp (bench.py pre-warms by time for the
    # same reason) — 20 ms of launches first
    ops.set_matmul_variant(rt, -1)
    import time
    t_end = time.perf_counter() + 0.02
    while time.perf_counter() < t_end:
        for _ in range(8):
            ops.matmul(rt, a, b, bias, out=c)
        rt.sync()
    line = f"{name:14s} {m:6d}x{n:5d}x{k:5d} {2.0 * m * n * k / 1e9:8.1f} GF |"
    for v in (int(x) for x in args.variants.split(",")):
        ops.set_matmul_variant(rt, v)
        for _ in range(5):
            ops.matmul(rt, a, b, bias, out=c)
        e0, e1 = Event(), Event()
        rt.record(e0)
        for _ in range(args.iters):
            ops.matmul(rt, a, b, bias, out=c)
        rt.record(e1)
        us = rt.elapsed_ms(e0, e1) / args.iters * 1e3
        line += f" {'heur' if v < 0 else names[v][:12]:>12s}: {us:7.1f} us {2.0 * m * n * k / us / 1e6:7.1f} TF |"
    print(line, flush=True)
ops.set_matmul_variant(rt, -1)
