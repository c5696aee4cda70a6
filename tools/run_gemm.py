"""Launch the headline GEMM a few times (for rocprofv3)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from infinitensor_amd import RocmRuntime, ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 2
tb = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
rt = RocmRuntime(0)
a = torch.randn(n, n, device="cuda").to(torch.bfloat16)
b = torch.randn(n, n, device="cuda").to(torch.bfloat16)
c = torch.empty(n, n, device="cuda", dtype=torch.bfloat16)
torch.cuda.synchronize()
ops.set_matmul_variant(rt, variant)
for _ in range(iters):
    ops.matmul(rt, a, b, None, False, tb, out=c)
rt.sync()
