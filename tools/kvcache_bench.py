"""Decode attention (AttentionKVCache) timing: B x H heads, n cached keys, head dim D; bytes = 2 n D sizeof(T) per head.
  python tools/kvcache_bench.py [--bh 32 --n 4096 --d 128 --dtype f16] [--splits 0,1,-1]   (-1 = the heuristic)"""
import argparse
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from infinitensor_amd import RocmRuntime, ops  # noqa: E402
from infinitensor_amd.runtime import Event  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bh", type=int, default=32)
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--d", type=int, default=128)
ap.add_argument("--dtype", default="f16")
ap.add_argument("--splits", default="0,1,-1")
a = ap.parse_args()
dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[a.dtype]
rt = RocmRuntime(0)
# several cache sets, rotated: 32 heads x 4096 keys x 128 x 2 B x 2 = 67 MB would otherwise sit in the 256 MiB Infinity Cache
sets = max(1, int(600e6 // (2 * a.bh * a.n * a.d * torch.empty(0, dtype=dt).element_size())))
kc = [torch.randn(1, a.bh, a.n, a.d, device="cuda").to(dt) for _ in range(sets)]
vc = [torch.randn(1, a.bh, a.n, a.d, device="cuda").to(dt) for _ in range(sets)]
q, k, v = (torch.randn(1, a.bh, 1, a.d, device="cuda").to(dt) for _ in range(3))
pos = torch.tensor([a.n - 1], dtype=torch.int32, device="cuda")
out = torch.empty_like(q)
torch.cuda.synchronize()
nbytes = 2.0 * a.bh * a.n * a.d * kc[0].element_size()
for sp in (int(x) for x in a.splits.split(",")):
    if sp < 0:
        os.environ.pop("IROCM_KVCACHE_SPLIT", None)
    else:
        os.environ["IROCM_KVCACHE_SPLIT"] = str(sp)
    for i in range(5):
        ops.attention_kvcache(rt, kc[i % sets], vc[i % sets], q, k, v, pos, out=out)
    # timed as a hipGraph of `iters` steps: a Python call costs 10-20 us, as much as the step itself
    iters = 40
    rt.sync()
    rt.begin_capture()
    for i in range(iters):
        ops.attention_kvcache(rt, kc[i % sets], vc[i % sets], q, k, v, pos, out=out)
    g = rt.end_capture()
    rt.launch_graph(g)
    e0, e1 = Event(), Event()
    rt.record(e0)
    rt.launch_graph(g)
    rt.record(e1)
    rt.sync()
    us = rt.elapsed_ms(e0, e1) / iters * 1e3
    print(f"bh {a.bh} n {a.n} d {a.d} {a.dtype} split {sp:>2}: {us:8.1f} us  {nbytes / us / 1e3:8.1f} GB/s  {nbytes / us / 1e3 / 8000:.3f} of 8 TB/s ({sets} rotating cache sets)")
