"""fp32 Conv2d timing on ResNet-50's distinct layers at batch 32 (the fp32 graph is what north_star's 1e-4 gate is defined on):
the fp32 matrix-instruction routes against the one-output-per-thread kernel; peak = 157.3 TF/s (v_mfma_f32_32x32x2_f32).
  python tools/conv32_bench.py [--batch 32] [--direct]"""
import argparse
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from conv_bench import RESNET50  # noqa: E402
from infinitensor_amd import RocmRuntime, ops  # noqa: E402
from infinitensor_amd.runtime import Event  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--direct", action="store_true", help="also time conv_direct32 (variant 1; slow)")
ap.add_argument("--forms", action="store_true", help="also time the implicit GEMM with 64^2 / 128^2 tiles forced (IROCM_CONV32_TILE) and "
                "the unit-stride pointwise layers as one fp32 GEMM per image (IROCM_CONV32_PW_BATCHED)")
a = ap.parse_args()
rt = RocmRuntime(0)
tot = {}
for cnt, c, h, f, r, st, pad in RESNET50:
    x = torch.randn(a.batch, c, h, h, device="cuda")
    w = torch.randn(f, c, r, r, device="cuda") / (c * r * r) ** 0.5
    b = torch.randn(f, device="cuda")
    oh = (h + 2 * pad - r) // st + 1
    y = torch.empty(a.batch, f, oh, oh, device="cuda")
    flop = 2.0 * a.batch * f * oh * oh * c * r * r
    torch.cuda.synchronize()
    line = f"x{cnt} C{c:<4d} {h:>3d}x{h:<3d} F{f:<4d} {r}x{r}/s{st} {flop / 1e9:7.2f} GF |"
    forms = (("mfma", -1, {}),)
    if a.forms:
        forms += (("t64", -1, {"IROCM_CONV32_TILE": "1"}), ("t128", -1, {"IROCM_CONV32_TILE": "2"}), ("pw-batched", -1, {"IROCM_CONV32_PW_BATCHED": "1"}),
                  ("s1", -1, {"IROCM_CONV32_SPLIT": "1"}), ("s2", -1, {"IROCM_CONV32_SPLIT": "2"}), ("s4", -1, {"IROCM_CONV32_SPLIT": "4"}))
    if a.direct:
        forms += (("direct", 1, {}),)
    for name, var, env in forms:
        for k in ("IROCM_CONV32_TILE", "IROCM_CONV32_PW_BATCHED", "IROCM_CONV32_SPLIT"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ops.set_conv_variant(rt, var)
        for _ in range(2):
            ops.conv2d(rt, x, w, pad, pad, st, st, bias=b, act=1, out=y)
        iters = 5 if var < 0 else 2
        e0, e1 = Event(), Event()
        rt.record(e0)
        for _ in range(iters):
            ops.conv2d(rt, x, w, pad, pad, st, st, bias=b, act=1, out=y)
        rt.record(e1)
        rt.sync()
        us = rt.elapsed_ms(e0, e1) / iters * 1e3
        tot[name] = tot.get(name, 0.0) + us * cnt
        if env:
            line += f" {name} {us:7.1f} us {flop / us / 1e6 / 157.3:.3f} |"
            continue
        line += f" {name} [{ops.conv_last_route(rt)}] {us:9.1f} us {flop / us / 1e6:7.1f} TF/s {flop / us / 1e6 / 157.3:.3f} of fp32 MFMA peak |"
    print(line, flush=True)
ops.set_conv_variant(rt, -1)
print("network conv total (us): " + "  ".join(f"{k}: {v:.0f}" for k, v in tot.items()))
