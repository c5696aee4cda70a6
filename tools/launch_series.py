"""Per-launch duration of 600 back-to-back 4096^3 bf16 GEMM launches from an idle chip, 25-launch window means (us): the clock ramp bench.py's warm-up has to cover."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from infinitensor_amd import RocmRuntime, ops
from infinitensor_amd.runtime import Event
rt = RocmRuntime(0)
g = torch.Generator(device="cuda").manual_seed(1234)
a = torch.randn(4096, 4096, device="cuda", generator=g).to(torch.bfloat16)
b = torch.randn(4096, 4096, device="cuda", generator=g).to(torch.bfloat16)
c = torch.empty(4096, 4096, device="cuda", dtype=torch.bfloat16)
torch.cuda.synchronize()
for trial in range(2):
    n = 600
    ev = [Event() for _ in range(n + 1)]
    rt.record(ev[0])
    for i in range(n):
        ops.matmul(rt, a, b, out=c)
        rt.record(ev[i + 1])
    rt.sync()
    d = [rt.elapsed_ms(ev[i], ev[i + 1]) * 1e3 for i in range(n)]
    print("trial", trial, " ".join(f"{sum(d[k:k+25])/25:.1f}" for k in range(0, n, 25)), flush=True)
    import time; time.sleep(0.5)
