"""Time the fused attention kernel on the BERT head shape (384 x 512 x 64, padding mask) and a long-sequence shape."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from infinitensor_amd import RocmRuntime, ops
from infinitensor_amd.runtime import Event
rt = RocmRuntime(0)
for bh, s, d, masked, causal in ((384, 512, 64, True, False), (384, 512, 64, False, False), (96, 2048, 64, False, False), (128, 512, 128, False, True)):
    q, k, v = (torch.randn(bh, s, d, device="cuda").half() for _ in range(3))
    m = torch.zeros(max(1, bh // 12), s, device="cuda").half() if masked else None
    o = torch.empty_like(q)
    torch.cuda.synchronize()
    for _ in range(5):
        ops.attention(rt, q, k, v, d ** -0.5, m, causal, out=o)
    e0, e1 = Event(), Event()
    rt.record(e0)
    for _ in range(20):
        ops.attention(rt, q, k, v, d ** -0.5, m, causal, out=o)
    rt.record(e1); rt.sync()
    us = rt.elapsed_ms(e0, e1) / 20 * 1e3
    fl = 4.0 * bh * s * s * d * (0.5 if causal else 1.0)
    ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float(), is_causal=causal) if not masked else None
    err = float((o.float() - ref).abs().max()) if ref is not None else float("nan")
    print(f"bh {bh} s {s} d {d} mask {masked} causal {causal}: {us:7.1f} us  {fl / us / 1e6:6.1f} TF/s  max err vs torch {err:.2e}")
