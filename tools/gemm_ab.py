"""Time the headline GEMM (bf16 4096^3, NN and NT) a few times in one process: for environment-variable A/B runs
(e.g. IROCM_GEMM_NT=1 python tools/gemm_ab.py).  Prints min / median of 5 batches of 100 launches."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from infinitensor_amd import RocmRuntime, ops
from infinitensor_amd.runtime import Event

rt = RocmRuntime(0)
n = 4096
a = torch.randn(n, n, device="cuda").to(torch.bfloat16)
b = torch.randn(n, n, device="cuda").to(torch.bfloat16)
c = torch.empty(n, n, device="cuda", dtype=torch.bfloat16)
torch.cuda.synchronize()
if "--check" in sys.argv:  # parity of THIS library's headline kernel vs fp32 torch.matmul of the same rounded operands (256 sampled rows)
    rows = torch.randint(0, n, (256,), device="cuda")
    for tb in (False, True):
        ops.matmul(rt, a, b, None, False, tb, out=c)
        rt.sync()
        want = a[rows].float() @ (b.float().t() if tb else b.float())
        err = (c[rows].float() - want).abs().max().item()
        print(f"check {'NT' if tb else 'NN'}: max abs err {err:.4f} (bf16 output, |c| ~ {want.abs().mean().item():.1f}) variant {rt.last_matmul_variant() if hasattr(rt, 'last_matmul_variant') else '?'}")
        assert err < 2.1, err  # |c| reaches ~300 at k = 4096: one bf16 ulp there is 2 (half an ulp of rounding + fp32 summation order)
for tb in (False, True):
    ts = []
    for _ in range(5):
        for _ in range(10):
            ops.matmul(rt, a, b, None, False, tb, out=c)
        e0, e1 = Event(), Event()
        rt.record(e0)
        for _ in range(100):
            ops.matmul(rt, a, b, None, False, tb, out=c)
        rt.record(e1)
        ts.append(rt.elapsed_ms(e0, e1) * 10)
    ts.sort()
    print(f"{'NT' if tb else 'NN'}: min {ts[0]:.1f} us ({2 * n ** 3 / ts[0] / 1e6:.0f} TF)  median {ts[2]:.1f} us ({2 * n ** 3 / ts[2] / 1e6:.0f} TF)")
