"""Fused attention timing (HIP events on the runtime stream): BERT / Llama head shapes."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from infinitensor_amd import RocmRuntime, ops
from infinitensor_amd.runtime import Event

rt = RocmRuntime(0)
for (bh, s, d, causal) in ((384, 512, 64, False), (128, 512, 128, False), (128, 512, 128, True), (32, 2048, 128, True), (32, 4096, 128, False), (96, 2048, 64, False)):
    q, k, v = (torch.randn(bh, s, d, device="cuda").half() for _ in range(3))
    o = torch.empty_like(q)
    torch.cuda.synchronize()
    for _ in range(3):
        ops.attention(rt, q, k, v, d ** -0.5, None, causal, out=o)
    e0, e1 = Event(), Event()
    rt.record(e0)
    for _ in range(20):
        ops.attention(rt, q, k, v, d ** -0.5, None, causal, out=o)
    rt.record(e1)
    rt.sync()
    ms = rt.elapsed_ms(e0, e1) / 20
    fl = 4.0 * bh * s * s * d * (0.5 if causal else 1.0)
    print(f"bh{bh} s{s} d{d} causal={int(causal)}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s", flush=True)
# BERT shape with its [B, S] padding mask
bh, s, d = 384, 512, 64
q, k, v = (torch.randn(bh, s, d, device="cuda").half() for _ in range(3))
m = torch.zeros(32, s, device="cuda").half()
o = torch.empty_like(q)
for _ in range(3):
    ops.attention(rt, q, k, v, d ** -0.5, m, False, out=o)
e0, e1 = Event(), Event()
rt.record(e0)
for _ in range(20):
    ops.attention(rt, q, k, v, d ** -0.5, m, False, out=o)
rt.record(e1)
rt.sync()
ms = rt.elapsed_ms(e0, e1) / 20
print(f"bh{bh} s{s} d{d} masked: {ms * 1e3:8.1f} us  {4.0 * bh * s * s * d / ms / 1e9:7.1f} TF/s", flush=True)
