"""Cycle ledger of ONE step (K-tile of 32) of the fp32 convolution on the matrix cores (csrc/gemm32.hip, conv_igemm32<1, true, TRACE>: the
64 x 64 tap-major form), four s_memtime stamps per step and wave:  python tools/conv32_timeline.py [--c 128 --h 28 --f 128 --batch 32]
  entry -> [counted wait: the step's weights (LDS-DMA) landed, own ds_writes drained] -> [s_barrier] -> [16 MFMAs + 8 gather requests
  + the next weights' DMA] -> [scatter: ds_write of the column requested a step ago; hipcc's own vmcnt wait for it sits here] -> next entry
Prints medians over all workgroups x steady steps per wave, in shader cycles (s_memtime), and how many workgroups the CU of workgroup 0 held."""
import argparse
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from infinitensor_amd import RocmRuntime, ops

ap = argparse.ArgumentParser()
ap.add_argument("--c", type=int, default=128)
ap.add_argument("--h", type=int, default=28)
ap.add_argument("--f", type=int, default=128)
ap.add_argument("--batch", type=int, default=32)
a = ap.parse_args()
rt = RocmRuntime(0)
n, c, h, f = a.batch, a.c, a.h, a.f
x = torch.randn(n, c, h, h, device="cuda")
w = torch.randn(f, c, 3, 3, device="cuda") / (c * 9) ** 0.5
b = torch.randn(f, device="cuda")
y = torch.empty(n, f, h, h, device="cuda")
tiles = -(-f // 64) * -(-(n * h * h) // 64)
trace = torch.zeros(tiles * 4 * 128, device="cuda", dtype=torch.int64)
torch.cuda.synchronize()
os.environ["IROCM_CONV32_TILE"] = "1"
os.environ["IROCM_CONV32_SPLIT"] = "1"
for _ in range(3):
    ops.conv2d(rt, x, w, 1, 1, 1, 1, bias=b, act=1, out=y)
rt.sync()
print("route", ops.conv_last_route(rt), "tiles", tiles, "steps per tile", 9 * c // 32)
os.environ["IROCM_CONV32_TRACE"] = hex(trace.data_ptr())
ops.conv2d(rt, x, w, 1, 1, 1, 1, bias=b, act=1, out=y)
rt.sync()
del os.environ["IROCM_CONV32_TRACE"]
t = trace.cpu().numpy().reshape(tiles, 4, 128)
names = ["counted wait (vmcnt(NE) + lgkmcnt(0))", "s_barrier", "16 MFMAs + 8 requests + DMA issue", "scatter (ds_writes; hipcc's vmcnt wait)"]
rows = []
for g in range(tiles):
    for wv in range(4):
        s = t[g, wv]
        s = s[s != 0]
        nst = (len(s) - 1) // 4
        if nst < 6:
            continue
        d = np.diff(s[: 4 * nst + 1]).reshape(nst, 4)
        rows.append(d[2:nst - 1])
d = np.concatenate(rows, 0)
med = np.median(d, 0)
print(f"{len(rows)} waves x steady steps; medians (shader cycles):")
for nm, v in zip(names, med):
    print(f"   {nm:<46s} {v:7.0f}")
print(f"   {'step':<46s} {med.sum():7.0f}   (MFMA issue floor: 16 x 64 = 1 024 per wave; waves per SIMD = resident workgroups per CU)")
# residency = sum of workgroup lifetimes / (launch duration x CUs). s_memtime counters are per XCD and not synchronised with each other, so
# the launch duration comes from HIP events of un-traced launches (cycles at the ~2.1 GHz the chip holds under this load: approximate).
from infinitensor_amd.runtime import Event
first = np.array([t[g, 0][t[g, 0] != 0][0] for g in range(tiles) if (t[g, 0] != 0).sum() > 8])
last = np.array([t[g, 0][t[g, 0] != 0][-1] for g in range(tiles) if (t[g, 0] != 0).sum() > 8])
life = float(np.median(last - first))
e0, e1 = Event(), Event()
rt.record(e0)
for _ in range(10):
    ops.conv2d(rt, x, w, 1, 1, 1, 1, bias=b, act=1, out=y)
rt.record(e1)
rt.sync()
us = rt.elapsed_ms(e0, e1) * 100.0
res = len(first) * life / (us * 2100.0) / 256
flop = 2.0 * n * f * h * h * c * 9
print(f"launch {us:.1f} us ({flop / us / 1e6:.1f} TF/s = {flop / us / 1e6 / 157.3:.2f} of the fp32 MFMA peak); median workgroup lifetime {life:.0f} cycles -> ~{res:.2f} workgroups "
      f"(= waves per SIMD) resident per CU on average; MFMA pipe demand per SIMD and step at that residency {res * 1024:.0f} of {med.sum():.0f} cycles = {res * 1024 / med.sum():.2f}")
