"""A fixed short attention workload for rocprofv3 (tools/profile_cmd.sh): BERT-base's head shape with its padding mask."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from infinitensor_amd import RocmRuntime, ops

rt = RocmRuntime(0)
bh, s, d = 384, 512, 64
q, k, v = (torch.randn(bh, s, d, device="cuda").half() for _ in range(3))
m = torch.zeros(32, s, device="cuda").half()
o = torch.empty_like(q)
torch.cuda.synchronize()
for _ in range(10):
    ops.attention(rt, q, k, v, d ** -0.5, m, False, out=o)
rt.sync()
