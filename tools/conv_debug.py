import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from infinitensor_amd import RocmRuntime, ops
n, c, h, f, r = map(int, sys.argv[1:6])
bias = len(sys.argv) > 6 and sys.argv[6] == "b"
rt = RocmRuntime(0)
x = torch.randn((n, c, h, h), device="cuda").half()
w = (torch.randn((f, c, r, r), device="cuda") / (c * r * r) ** 0.5).half()
b = torch.randn((f,), device="cuda").half() if bias else None
torch.cuda.synchronize()
ops.set_conv_variant(rt, 2)
y = ops.conv2d(rt, x, w, r // 2, r // 2, bias=b, act=1 if bias else 0)
rt.sync(); torch.cuda.synchronize()
ops.set_conv_variant(rt, 1)
yg = ops.conv2d(rt, x, w, r // 2, r // 2, bias=b, act=1 if bias else 0)
rt.sync()
print(sys.argv[1:], "maxdiff", (y.float() - yg.float()).abs().max().item())
