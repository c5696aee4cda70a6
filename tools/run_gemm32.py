"""Launch the fp32 4096^3 MatMul a few times (for rocprofv3 / tools/profile_cmd.sh)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from infinitensor_amd import RocmRuntime, ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rt = RocmRuntime(0)
a = torch.randn(n, n, device="cuda")
b = torch.randn(n, n, device="cuda")
c = torch.empty(n, n, device="cuda")
torch.cuda.synchronize()
for _ in range(iters):
    ops.matmul(rt, a, b, out=c)
rt.sync()
print(ops.matmul_last_variant(rt))
