"""Depthwise Conv2d timing on EfficientNet-Lite4's depthwise layers (input 300 x 300, batch 32; docs/SUPPORT_MATRIX_CN.md:24-27 lists
the model as validated): algorithmic bytes = input + output once; generic implicit GEMM (variant 1) beside the depthwise kernel.
  python tools/dwconv_bench.py [--batch 32] [--dtype f16]"""
import argparse
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from infinitensor_amd import RocmRuntime, ops  # noqa: E402
from infinitensor_amd.runtime import Event  # noqa: E402

# (C, H, k, stride) of the depthwise layer of every MBConv stage of EfficientNet-Lite4 (width 1.4, depth 1.8, input 300; expanded channels)
LAYERS = [(32, 150, 3, 1), (144, 150, 3, 2), (192, 75, 3, 1), (192, 75, 5, 2), (336, 38, 5, 1), (336, 38, 3, 2), (672, 19, 3, 1),
          (672, 19, 5, 1), (960, 19, 5, 1), (960, 19, 5, 2), (1632, 10, 5, 1), (1632, 10, 3, 1), (2688, 10, 3, 1)]

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--dtype", default="f16")
ap.add_argument("--generic", action="store_true", help="also time the generic implicit GEMM (slow)")
ap.add_argument("--layers", default="", help="comma-separated indices into the layer table (default: all)")
ap.add_argument("--th-mults", default="", help="comma-separated multiples of the kernel height: also time the depthwise kernel with that "
                "many output rows per thread (IROCM_DW_TH), e.g. 1,2,3,4")
a = ap.parse_args()
dt = {"f16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
rt = RocmRuntime(0)
tot = {"dw": 0.0, "floor": 0.0}
for c, h, k, st in ([LAYERS[int(i)] for i in a.layers.split(",")] if a.layers else LAYERS):
    pad = k // 2
    oh = (h + 2 * pad - k) // st + 1
    x = torch.randn(a.batch, c, h, h, device="cuda").to(dt)
    w = (torch.randn(c, 1, k, k, device="cuda") / k).to(dt)
    b = torch.randn(c, device="cuda").to(dt)
    y = torch.empty(a.batch, c, oh, oh, device="cuda", dtype=dt)
    torch.cuda.synchronize()
    nbytes = 2.0 * a.batch * c * (h * h + oh * oh)
    line = f"C{c:<5d} {h:>3d}x{h:<3d} {k}x{k}/s{st} {nbytes / 1e6:7.1f} MB |"
    forms = (("dw", -1, 0),) + tuple((f"th{int(m) * k}", -1, int(m) * k) for m in a.th_mults.split(",") if m) + ((("generic", 1, 0),) if a.generic else ())
    for name, var, th in forms:
        if th:
            os.environ["IROCM_DW_TH"] = str(th)
        else:
            os.environ.pop("IROCM_DW_TH", None)
        ops.set_conv_variant(rt, var)
        for _ in range(3):
            ops.conv2d(rt, x, w, pad, pad, st, st, bias=b, act=1, out=y)
        # timed as a hipGraph of `iters` launches: a Python call costs 10-20 us, as much as the shorter kernels
        iters = 20 if var < 0 else 3
        rt.sync()
        rt.begin_capture()
        for _ in range(iters):
            ops.conv2d(rt, x, w, pad, pad, st, st, bias=b, act=1, out=y)
        g = rt.end_capture()
        rt.launch_graph(g)
        e0, e1 = Event(), Event()
        rt.record(e0)
        rt.launch_graph(g)
        rt.record(e1)
        rt.sync()
        us = rt.elapsed_ms(e0, e1) / iters * 1e3
        if name == "dw":
            tot["dw"] += us
            tot["floor"] += nbytes / 8e12 * 1e6
        if th:
            line += f" {name} {us:6.1f} |"
            continue
        line += f" {name} [{ops.conv_last_route(rt)}] {us:8.1f} us {nbytes / us / 1e3:7.0f} GB/s {nbytes / us / 1e3 / 8000:.3f} of 8 TB/s |"
    print(line, flush=True)
ops.set_conv_variant(rt, -1)
print(f"all depthwise layers: {tot['dw']:.1f} us; at 8 TB/s: {tot['floor']:.1f} us ({tot['floor'] / tot['dw']:.3f})")
