"""One fp32 conv layer, a few launches (for rocprofv3): python tools/run_conv32.py C H F R stride pad [batch]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from infinitensor_amd import RocmRuntime, ops
c, h, f, r, st, pad = (int(v) for v in sys.argv[1:7])
n = int(sys.argv[7]) if len(sys.argv) > 7 else 32
rt = RocmRuntime(0)
x = torch.randn(n, c, h, h, device="cuda"); w = torch.randn(f, c, r, r, device="cuda") / (c * r * r) ** 0.5; b = torch.randn(f, device="cuda")
torch.cuda.synchronize()
for _ in range(5):
    y = ops.conv2d(rt, x, w, pad, pad, st, st, bias=b, act=1)
rt.sync()
print(ops.conv_last_route(rt))
