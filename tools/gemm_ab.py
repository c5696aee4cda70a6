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
