"""Cycle ledger of ONE K-tile of the persistent GEMM (csrc/gemm256p_kernel.h, TRACE build with six stamps per K-tile): what fills the
cycles beside the 2 048 cycles of MFMA issue.  python tools/gemm_ktile_ledger.py [--m 4096 --n 4096 --k 1024]
Per wave and K-tile the stamps are: L1 start | C1 start (behind L1's barrier) | after C1's MFMA burst was issued | L2 start (behind
C1's barrier) | C2 start (behind L2's barrier) | after C2's burst. Intervals: L1 = reads + A DMA + wait + barrier; C1 issue = 32 MFMAs
issued (16 cycles each when the pipe is free); C1 tail = cursor work + the barrier (waits for the partner row's L phase); L2, C2 alike.
Printed: the median over the steady K-tiles (first two and last one of a tile dropped) of wave 0 (row 0) and wave 4 (row 1) over all
workgroups, in s_memtime ticks = shader cycles (guide: the stamps themselves add ~11 %)."""
import argparse
import ctypes as C
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from infinitensor_amd import RocmRuntime, lib
from infinitensor_amd._lib import check

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=4096)
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--k", type=int, default=1024)
ap.add_argument("--fine", type=int, default=1, help="1: six stamps per K-tile; 4: ten (two more inside each LOAD phase: everything issued | data landed)")
a_ = ap.parse_args()
os.environ["IROCM_GEMM_TRACE_FINE"] = str(a_.fine)
rt = RocmRuntime(0)
a = torch.randn(a_.m, a_.k, device="cuda").to(torch.bfloat16)
b = torch.randn(a_.k, a_.n, device="cuda").to(torch.bfloat16)
c = torch.empty(a_.m, a_.n, device="cuda", dtype=torch.bfloat16)
cus = rt.device_info()["compute_units"]
tiles = -(-a_.m // 256) * -(-a_.n // 256)
grid = min(tiles, cus)
trace = torch.zeros(grid * 8 * 128, device="cuda", dtype=torch.int64)
torch.cuda.synchronize()
for _ in range(3):
    check(lib().infini_rocm_probe_gemm_timeline(rt.handle, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(c.data_ptr()),
                                                a_.m, a_.n, a_.k, 256, C.c_void_p(trace.data_ptr())))
rt.sync()
t = trace.cpu().numpy().reshape(grid, 8, 128)
NS = 6 if a_.fine == 1 else 10  # stamps per K-tile
nk = min(a_.k // 64, 20 if NS == 6 else 12)  # (the trace strip holds 128 stamps per wave: the first 20 / 12 K-tiles of a longer K)
names = ["L1 (reads, A DMA, wait, barrier)", "C1 issue (32 MFMAs)", "C1 tail (cursors, barrier)", "L2 (reads, B DMA, waits, barrier)",
         "C2 issue (32 MFMAs)", "C2 tail (barrier)"]
if NS == 10:
    names = ["L1 issue (24 reads + 4 A pieces)", "L1 lgkm wait (+ pieces: variant 2)", "L1 barrier", "C1 issue (32 MFMAs)", "C1 tail (cursors, barrier)",
             "L2 issue (8 reads + 4 B pieces)", "L2 vmcnt + lgkm wait", "L2 barrier", "C2 issue (32 MFMAs)", "C2 tail (barrier)"]
for wv in (0, 4):
    rows = []
    for g in range(grid):
        s = t[g, wv]
        s = s[s != 0]
        if len(s) < 1 + NS * nk:
            continue
        kt = s[1:1 + NS * nk + 1] if len(s) > 1 + NS * nk else None
        if kt is None:
            continue
        d = np.diff(kt).reshape(-1)[: NS * nk].reshape(nk, NS) if len(kt) == NS * nk + 1 else None
        if d is None:
            continue
        rows.append(d[2:nk - 1])
    d = np.concatenate(rows, 0)
    med = np.median(d, 0)
    print(f"wave {wv} (row {wv // 4}): {len(rows)} workgroups x {nk - 3} steady K-tiles")
    for nm, v in zip(names, med):
        print(f"   {nm:<40s} {v:7.0f}")
    print(f"   {'K-tile':<40s} {med.sum():7.0f}   (MFMA issue floor 2 x 512 = 1 024 per row ... 2 048 per K-tile and SIMD shared by the two rows)")
