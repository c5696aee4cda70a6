"""Where inside the 256 x 256 tile does a (variant) library's headline GEMM differ from fp32 torch.matmul? Prints, for bf16 4096^3 NN, the
share of wrong elements per 16-row block x 16-column block of the tile (aggregated over all tiles) and a few sample values.
INFINI_ROCM_LIB=... python tools/gemm_diag.py [--k 4096]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from infinitensor_amd import RocmRuntime, ops

ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, default=4096)
ap.add_argument("--n", type=int, default=4096)
a_ = ap.parse_args()
rt = RocmRuntime(0)
n, k = a_.n, a_.k
torch.manual_seed(0)
a = torch.randn(n, k, device="cuda").to(torch.bfloat16)
b = torch.randn(k, n, device="cuda").to(torch.bfloat16)
c = torch.empty(n, n, device="cuda", dtype=torch.bfloat16)
ops.matmul(rt, a, b, None, False, False, out=c)
rt.sync()
want = a.float() @ b.float()
err = (c.float() - want).abs()
tol = 0.02 * want.abs() + 0.5
bad = err > tol
print(f"k {k}: wrong elements {bad.float().mean().item():.4f}; max err {err.max().item():.2f}")
blk = bad.view(n // 256, 16, 16, n // 256, 16, 16).float().mean(dim=(0, 2, 3, 5))  # [row block of tile][col block of tile]
torch.set_printoptions(linewidth=250, precision=2, sci_mode=False)
print("share wrong per (16-row block, 16-col block) of the 256 x 256 tile:")
print(blk.cpu())
tiles = bad.view(n // 256, 256, n // 256, 256).float().mean(dim=(1, 3))
print("share wrong per tile (first 8 x 8 tiles):")
print(tiles[:8, :8].cpu())
idx = bad.nonzero()[:6]
for i, j in idx.tolist():
    print(f"  C[{i}][{j}] = {c[i, j].item():.2f}, want {want[i, j].item():.2f}")
# is the wrong value a partial sum? compare against prefix sums over K-tiles of 64 for the first wrong element
if len(idx):
    i, j = idx[0].tolist()
    pref = torch.cumsum((a[i].float().view(-1, 64) * b[:, j].float().view(-1, 64)).sum(1), 0)
    d = (pref - c[i, j].float()).abs()
    print("  nearest prefix sum over K-tiles:", int(d.argmin().item()) + 1, "of", k // 64, "K-tiles, diff", d.min().item())
    tot = pref[-1]
    print("  want - got =", (tot - c[i, j].float()).item())

# the TRACE instantiation (tools/gemm_ktile_ledger.py's kernel) on the same operands: its own register allocation
import ctypes as C
from infinitensor_amd import lib
from infinitensor_amd._lib import check
cus = rt.device_info()["compute_units"]
grid = min((n // 256) ** 2, cus)
trace = torch.zeros(grid * 8 * 128, device="cuda", dtype=torch.int64)
c2 = torch.zeros(n, n, device="cuda", dtype=torch.bfloat16)
check(lib().infini_rocm_probe_gemm_timeline(rt.handle, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(c2.data_ptr()), n, n, k, 256,
                                            C.c_void_p(trace.data_ptr())))
rt.sync()
err2 = (c2.float() - want).abs()
print(f"TRACE build: wrong elements {(err2 > tol).float().mean().item():.4f}; max err {err2.max().item():.2f}")
