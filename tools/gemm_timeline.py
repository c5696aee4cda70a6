"""Per-wave phase timeline of the persistent GEMM (csrc/gemm256p_kernel.h, TRACE build): where the cycles of a tile
boundary go.  python tools/gemm_timeline.py [--m 16384 --n 3072 --k 768 --tile 256] [--wg 0,100]
Prints, for the chosen workgroups and waves 0 (row 0) and 4 (row 1, same SIMD): cycles per K-tile, the epilogue, the
first K-tiles after a boundary, the final drain (s_memtime ticks; 100 MHz constant clock -> 10 ns each)."""
import argparse
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from infinitensor_amd import RocmRuntime, lib
from infinitensor_amd._lib import check

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=16384)
ap.add_argument("--n", type=int, default=3072)
ap.add_argument("--k", type=int, default=768)
ap.add_argument("--tile", type=int, default=256)
ap.add_argument("--wg", default="0,100")
a_ = ap.parse_args()
rt = RocmRuntime(0)
a = torch.randn(a_.m, a_.k, device="cuda").to(torch.bfloat16)
b = (torch.randn(a_.k, a_.n, device="cuda") * 0.05).to(torch.bfloat16)
c = torch.empty(a_.m, a_.n, device="cuda", dtype=torch.bfloat16)
cus = rt.device_info()["compute_units"]
tiles = -(-a_.m // 256) * -(-a_.n // a_.tile)
grid = min(tiles, cus)
trace = torch.zeros(grid * 8 * 128, device="cuda", dtype=torch.int64)
torch.cuda.synchronize()
for _ in range(3):
    check(lib().infini_rocm_probe_gemm_timeline(rt.handle, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(c.data_ptr()),
                                                a_.m, a_.n, a_.k, a_.tile, C.c_void_p(trace.data_ptr())))
rt.sync()
t = trace.cpu().numpy().reshape(grid, 8, 128)
ref = (a.float() @ b.float())
print("max rel err vs torch:", float(((c.float() - ref).abs() / (ref.abs() + 1)).max()))
nk = a_.k // 64
my = -(-(tiles - 0) // grid)
print(f"tiles {tiles}, grid {grid}, nk {nk}, tiles per workgroup <= {my}")
t0 = t[:, :, 0].min()
for wg in (int(x) for x in a_.wg.split(",")):
    for w in (0, 4):
        s = t[wg, w]
        n = int((s != 0).sum())
        s = s[:n] - t0
        print(f"-- wg {wg} wave {w}: {n} stamps, kernel entry at +{s[0]} ticks, end at +{s[-1]}")
        i = 1
        line = []
        for tile in range(my):
            if i + 2 * nk + 16 > n:
                break
            kt = s[i:i + 2 * nk:2]
            l2 = s[i + 1:i + 2 * nk:2]
            # stamps behind the K loop: e0 | row barrier | epilogue entry | bias | 16-byte path entered | address set-up | first pack + exchange | after store step 0 .. 7 | end
            e0, e1 = s[i + 2 * nk], s[i + 2 * nk + 15]
            pre = np.diff(s[i + 2 * nk:i + 2 * nk + 7])
            steps = np.diff(s[i + 2 * nk + 6:i + 2 * nk + 15])
            d = np.diff(np.append(kt, e0))
            print(f"   tile {tile}: first L1 at +{kt[0]}, K-tile ticks {d.tolist()}, epilogue {e1 - e0} (row barrier / entry / bias / path / set-up / first exchange {pre.tolist()}, store steps {steps.tolist()})")
            i += 2 * nk + 16
        print(f"   final drain (last epilogue end -> all stores done): {s[-1] - s[-2]}")
