"""Per-wave phase timeline of the conv tap GEMM (csrc/gemm256p_kernel.h, CONV = 3, TRACE build): where the time of a 3 x 3 layer goes.
  python tools/conv_tap_timeline.py [--c 256 --h 14 --f 256 --stride 1 --batch 128] [--wg 0,8,100] [--split 0]
Stamps (s_memtime, 100 MHz: 10 ns each): kernel entry; per K-tile L1 and L2; tile end; row barrier; [split-K: exchange entry, payload
drained, per source: flag seen, row blocks added]; after the epilogue."""
import argparse
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from infinitensor_amd import RocmRuntime, ops

ap = argparse.ArgumentParser()
ap.add_argument("--c", type=int, default=256)
ap.add_argument("--h", type=int, default=14)
ap.add_argument("--f", type=int, default=256)
ap.add_argument("--stride", type=int, default=1)
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--split", type=int, default=0, help="force the split factor (1 = none; 0 = the launcher's choice)")
ap.add_argument("--wg", default="0,8,16,100")
a = ap.parse_args()
rt = RocmRuntime(0)
n, c, h, f, st = a.batch, a.c, a.h, a.f, a.stride
xbuf = torch.empty((n * c * h * h + 64 + 256,), device="cuda", dtype=torch.float16)
x = xbuf[256: 256 + n * c * h * h].view(n, c, h, h).copy_(torch.randn((n, c, h, h), device="cuda"))
w = (torch.randn((f, c, 3, 3), device="cuda") / (c * 9) ** 0.5).to(torch.float16)
b = torch.randn((f,), device="cuda").to(torch.float16)
oh = (h + st - 1) // st
y = torch.empty((n, f, oh, oh), device="cuda", dtype=torch.float16)
grid = 256
trace = torch.zeros(grid * 8 * 128, device="cuda", dtype=torch.int64)
torch.cuda.synchronize()
if a.split:
    os.environ["IROCM_CONV_TAP_SPLIT"] = str(a.split)
os.environ["IROCM_CONV_TAP_NT"] = "4" if a.split == 1 else ""
if not os.environ["IROCM_CONV_TAP_NT"]:
    del os.environ["IROCM_CONV_TAP_NT"]
ops.set_conv_variant(rt, 7)
for _ in range(3):
    ops.conv2d(rt, x, w, 1, 1, st, st, bias=b, act=1, out=y)
rt.sync()
os.environ["IROCM_CONV_TAP_TRACE"] = hex(trace.data_ptr())
ops.conv2d(rt, x, w, 1, 1, st, st, bias=b, act=1, out=y)
rt.sync()
print("route", ops.conv_last_route(rt))
t = trace.cpu().numpy().reshape(grid, 8, 128)
live = t[:, :, 0] != 0
t0 = t[:, :, 0][live].min()
ends = np.array([t[g, wv][t[g, wv] != 0][-1] for g in range(grid) for wv in range(8) if live[g, wv]])
print(f"workgroups traced {int(live[:, 0].sum())}; kernel entry spread {int(t[:, :, 0][live].max() - t0)} ticks; last stamp at +{int(ends.max() - t0)} ticks (x 10 ns)")
for wg in (int(v) for v in a.wg.split(",")):
    for wv in (0, 4):
        s = t[wg, wv]
        k = int((s != 0).sum())
        if k == 0:
            print(f"-- wg {wg} wave {wv}: not traced")
            continue
        s = s[:k] - t0
        d = np.diff(s)
        print(f"-- wg {wg} wave {wv}: {k} stamps, entry +{int(s[0])}, end +{int(s[-1])}; intervals: {d.tolist()}")
