"""A few launches of the fused stem (Conv 7 x 7 / 2 + bias + ReLU + MaxPool 3 x 3 / 2) at batch 128 for tools/profile_cmd.sh."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from infinitensor_amd import RocmRuntime, ops  # noqa: E402

rt = RocmRuntime(0)
rt.use_torch_stream()
x = torch.rand(128, 3, 224, 224, device="cuda").half()
w = (torch.randn(64, 3, 7, 7, device="cuda") * 0.1).half()
b = torch.randn(64, device="cuda").half()
y = torch.empty(128, 64, 56, 56, device="cuda", dtype=torch.float16)
ops.set_conv_const_weights(rt, True)
for _ in range(6):
    ops.conv2d_pool(rt, x, w, b, 3, 3, 2, 2, 3, 2, 1, out=y)
rt.sync()
from infinitensor_amd.runtime import Event  # noqa: E402

e0, e1 = Event(), Event()
rt.record(e0)
for _ in range(50):
    ops.conv2d_pool(rt, x, w, b, 3, 3, 2, 2, 3, 2, 1, out=y)
rt.record(e1)
rt.sync()
print(f"stem+pool fused bs128: {rt.elapsed_ms(e0, e1) / 50 * 1e3:.1f} us")
