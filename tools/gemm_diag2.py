"""TRACE-build correctness probe for K-loop variants: bf16 NN 1024 x 1024 x k through infini_rocm_probe_gemm_timeline (always the persistent
256 x 256 kernel), share of wrong elements per 16-row block of the tile and per K."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from infinitensor_amd import RocmRuntime, lib
from infinitensor_amd._lib import check

rt = RocmRuntime(0)
n = 1024
torch.set_printoptions(linewidth=250, precision=2, sci_mode=False)
for k in (64, 128, 192, 256, 1024):
    torch.manual_seed(k)
    a = torch.randn(n, k, device="cuda").to(torch.bfloat16)
    b = torch.randn(k, n, device="cuda").to(torch.bfloat16)
    c = torch.zeros(n, n, device="cuda", dtype=torch.bfloat16)
    trace = torch.zeros(16 * 8 * 128, device="cuda", dtype=torch.int64)
    check(lib().infini_rocm_probe_gemm_timeline(rt.handle, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(c.data_ptr()), n, n, k, 256,
                                                C.c_void_p(trace.data_ptr())))
    rt.sync()
    want = a.float() @ b.float()
    err = (c.float() - want).abs()
    bad = ~(err <= 0.02 * want.abs() + 0.3)
    print(f"k {k}: wrong {bad.float().mean().item():.4f} nan {torch.isnan(c.float()).float().mean().item():.4f}")
    if bad.any():
        blk = bad.view(n // 256, 16, 16, n // 256, 16, 16).float().mean(dim=(0, 2, 3, 5))
        print(" per row block:", blk.mean(1).cpu().tolist())
        print(" per col block:", blk.mean(0).cpu().tolist())
        i, j = bad.nonzero()[0].tolist()
        # which partial sums does the wrong value match? per-K-tile contributions
        contrib = (a[i].float().view(-1, 32) * b[:, j].float().view(-1, 32)).sum(1)  # per 32-deep k-step
        print(f" C[{i}][{j}] = {c[i, j].item():.3f} want {want[i, j].item():.3f}; k-step contributions {contrib.cpu().tolist()[:8]}")
