"""Per-layer Conv2d timing on the GPU (ResNet-50 bs128 layer shapes), per kernel variant.

  python tools/conv_bench.py [--batch 128] [--dtype f16] [--variants -1,1,2,3] [--iters 10]
Prints one line per distinct layer: shape, GFLOP, and for each variant ms and TFLOP/s (HIP events on the
runtime's stream), then the total over the whole network weighted by layer multiplicity.
"""
from __future__ import annotations

import argparse
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from infinitensor_amd import RocmRuntime, ops  # noqa: E402
from infinitensor_amd.runtime import Event  # noqa: E402

# (count, C, H, F, R, stride, pad)
RESNET50 = [
    (1, 3, 224, 64, 7, 2, 3),
    # stage 1 (56x56)
    (1, 64, 56, 64, 1, 1, 0), (3, 64, 56, 64, 3, 1, 1), (4, 64, 56, 256, 1, 1, 0), (2, 256, 56, 64, 1, 1, 0),
    # stage 2
    (1, 256, 56, 128, 1, 1, 0), (1, 128, 56, 128, 3, 2, 1), (4, 128, 28, 512, 1, 1, 0), (1, 256, 56, 512, 1, 2, 0),
    (3, 512, 28, 128, 1, 1, 0), (3, 128, 28, 128, 3, 1, 1),
    # stage 3
    (1, 512, 28, 256, 1, 1, 0), (1, 256, 28, 256, 3, 2, 1), (6, 256, 14, 1024, 1, 1, 0), (1, 512, 28, 1024, 1, 2, 0),
    (5, 1024, 14, 256, 1, 1, 0), (5, 256, 14, 256, 3, 1, 1),
    # stage 4
    (1, 1024, 14, 512, 1, 1, 0), (1, 512, 14, 512, 3, 2, 1), (3, 512, 7, 2048, 1, 1, 0), (1, 1024, 14, 2048, 1, 2, 0),
    (2, 2048, 7, 512, 1, 1, 0), (2, 512, 7, 512, 3, 1, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--variants", default="-1,1,2,3")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--res", action="store_true", help="add a residual operand to the expanding pointwise layers (f > c, 1 x 1 / 1): the join of a bottleneck")
    ap.add_argument("--tap-nt", default="", help="for variant 7: comma-separated tile widths (2,3,4) to force, one column each")
    ap.add_argument("--layers", default="", help="comma-separated indices into the layer table (default: all)")
    args = ap.parse_args()
    dt = {"f16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
    rt = RocmRuntime(0)
    variants = [int(v) for v in args.variants.split(",")]
    totals = {}
    tot_flop = 0.0
    tot_floor = 0.0
    table = [RESNET50[int(i)] for i in args.layers.split(",")] if args.layers else RESNET50
    for cnt, c, h, f, r, st, pad in table:
        # (64 spare elements behind the input, as in the plugin's arena: the pixel-slot GEMM reads up to 14 bytes past a ragged plane)
        # and 256 in front: the tap mode (3 x 3 layers as one GEMM) reads up to one row + one pixel in front of the first plane)
        xbuf = torch.empty((args.batch * c * h * h + 64 + 256,), device="cuda", dtype=dt)
        x = xbuf[256: 256 + args.batch * c * h * h].view(args.batch, c, h, h).copy_(torch.randn((args.batch, c, h, h), device="cuda"))
        w = (torch.randn((f, c, r, r), device="cuda") / (c * r * r) ** 0.5).to(dt)
        b = torch.randn((f,), device="cuda").to(dt)
        oh = (h + 2 * pad - r) // st + 1
        y = torch.empty((args.batch, f, oh, oh), device="cuda", dtype=dt)
        res = torch.randn((args.batch, f, oh, oh), device="cuda").to(dt) if (args.res and r == 1 and st == 1 and f > c) else None
        flop = 2.0 * args.batch * f * oh * oh * c * r * r
        tot_flop += flop * cnt
        # roofline floor of the layer: algorithmic bytes (input + weights + output, each once) at the 6.3 TB/s the guide
        # measures as achievable, vs the FLOPs at the 2.15 PF the chip sustains at its ~2.05 GHz load clock
        nbytes = 2.0 * (args.batch * c * h * h + f * c * r * r + args.batch * f * oh * oh * (2 if res is not None else 1))
        floor_us = max(nbytes / 6.3e12, flop / 2.15e15) * 1e6
        tot_floor += floor_us * cnt
        line = f"x{cnt} C{c:<4d} {h:>3d}x{h:<3d} F{f:<4d} {r}x{r}/s{st} {flop / 1e9:8.2f} GF {nbytes / 1e6:7.1f} MB floor {floor_us:6.1f} us ({'hbm' if nbytes / 6.3e12 > flop / 2.15e15 else 'mfma'}) |"
        if os.environ.get("CONV_BENCH_PTRS"):
            print(f"x {x.data_ptr():#x}+{x.numel() * 2:#x} w {w.data_ptr():#x} b {b.data_ptr():#x} y {y.data_ptr():#x}+{y.numel() * 2:#x} "
                  f"ws {rt.workspace(1):#x}", flush=True)
        torch.cuda.synchronize()
        runs = []
        for v in variants:
            if v == 7 and args.tap_nt:
                runs += [(v, nt) for nt in args.tap_nt.split(",")]
            else:
                runs.append((v, None))
        for v, tnt in runs:
            ops.set_conv_variant(rt, v)
            if tnt is not None:
                os.environ["IROCM_CONV_TAP_NT"] = tnt
            else:
                os.environ.pop("IROCM_CONV_TAP_NT", None)
            for _ in range(2):
                ops.conv2d(rt, x, w, pad, pad, st, st, bias=b, act=1, out=y, residual=res)
            e0, e1 = Event(), Event()
            rt.record(e0)
            for _ in range(args.iters):
                ops.conv2d(rt, x, w, pad, pad, st, st, bias=b, act=1, out=y, residual=res)
            rt.record(e1)
            rt.sync()
            ms = rt.elapsed_ms(e0, e1) / args.iters
            key = v if tnt is None else f"{v}/nt{tnt}"
            totals[key] = totals.get(key, 0.0) + ms * cnt
            route = {"tap_gemm": "X", "tap_gemm_splitk": "K", "pixel_gemm": "P", "tap_shifted": "T", "resident": "R", "batched_gemm": "B", "generic": "G", "direct32": "D"}[ops.conv_last_route(rt)]
            line += f" v{key}: {ms * 1e3:8.1f} us {flop / ms / 1e9:7.1f} TF x{ms * 1e3 / floor_us:4.1f} {route} |"
        print(line, flush=True)
    ops.set_conv_variant(rt, -1)
    print("route letters: X 3 x 3 layer as one GEMM with K = 9 C (tap mode), K the same with split-K, P pixel-slot GEMM on the persistent kernels, R resident-weights kernel, T tap-shifted / patch kernels (conv_s1.hip), B batched GEMM, G generic")
    print(f"network roofline floor: {tot_floor / 1e3:.3f} ms")
    print("network conv total: " + "  ".join(f"v{v}: {t:.3f} ms ({tot_flop / t / 1e9:.1f} TF/s)" for v, t in totals.items()))


if __name__ == "__main__":
    main()
