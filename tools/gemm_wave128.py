"""The four-wave GEMM (csrc/gemm128w.hip, variant "wave128") beside the shipped persistent kernel: parity against an fp64 product of the
rounded operands on small whole-tile shapes in all four layouts, then TFLOP/s of both kernels interleaved on the large shapes.
  python tools/gemm_wave128.py [--reps 20] [--no-check] [--ab] [--clock]
--ab: the routing evidence (profiles/r06_gemm_wave128_ab.txt) — shipped heuristic against the forced four-wave kernel, eight interleaved
rounds of 30 launches per shape, min / median / max. --clock (run with IROCM_W128_DBG=8): the core clock the K loops ran at."""
import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from infinitensor_amd import RocmRuntime, ops
from infinitensor_amd.runtime import Event


def check(rt, dt, b, m, n, k, ta, tb, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn((b, k, m) if ta else (b, m, k), device="cuda", generator=g).to(dt)
    bb = torch.randn((b, n, k) if tb else (b, k, n), device="cuda", generator=g).to(dt)
    if b == 1:
        a, bb = a[0], bb[0]
    ops.set_matmul_variant(rt, ops.matmul_variants().index("wave128"))
    try:
        c = ops.matmul(rt, a, bb, trans_a=ta, trans_b=tb)
        used = ops.matmul_last_variant(rt)
    finally:
        ops.set_matmul_variant(rt, -1)
    rt.sync()
    a64 = a.double().transpose(-1, -2) if ta else a.double()
    b64 = bb.double().transpose(-1, -2) if tb else bb.double()
    want = a64 @ b64
    err = (c.double() - want).abs()
    tol = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    bound = tol * want.abs() + 2.0 ** -17 * (a64.abs() @ b64.abs())
    bad = int((err > bound).sum().item())
    return {"shape": [b, m, n, k], "ta": ta, "tb": tb, "dtype": str(dt).split(".")[-1], "variant": used, "max_err": float(err.max().item()),
            "bad": bad}


def timeit(rt, dt, m, n, k, tb, variant, reps):
    a = torch.randn(m, k, device="cuda").to(dt)
    b = torch.randn((n, k) if tb else (k, n), device="cuda").to(dt)
    out = torch.empty(m, n, device="cuda", dtype=dt)
    ops.set_matmul_variant(rt, variant)
    try:
        for _ in range(3):
            ops.matmul(rt, a, b, trans_b=tb, out=out)
        e0, e1 = Event(), Event()
        rt.record(e0)
        for _ in range(reps):
            ops.matmul(rt, a, b, trans_b=tb, out=out)
        rt.record(e1)
        ms = rt.elapsed_ms(e0, e1) / reps
        used = ops.matmul_last_variant(rt)
    finally:
        ops.set_matmul_variant(rt, -1)
    return round(2.0 * m * n * k / ms / 1e9, 1), used


def ab(rt, w, dt, m, n, k, ta, tb, rounds=8, reps=30):
    import statistics

    a = torch.randn((k, m) if ta else (m, k), device="cuda").to(dt)
    b = torch.randn((n, k) if tb else (k, n), device="cuda").to(dt)
    out = torch.empty(m, n, device="cuda", dtype=dt)
    res = {"persist": [], "wave128": []}
    try:
        for _ in range(rounds):
            for name, v in (("persist", 4), ("wave128", w)):
                ops.set_matmul_variant(rt, v)
                for _ in range(3):
                    ops.matmul(rt, a, b, trans_a=ta, trans_b=tb, out=out)
                e0, e1 = Event(), Event()
                rt.record(e0)
                for _ in range(reps):
                    ops.matmul(rt, a, b, trans_a=ta, trans_b=tb, out=out)
                rt.record(e1)
                res[name].append(round(2.0 * m * n * k / (rt.elapsed_ms(e0, e1) / reps) / 1e9, 1))
    finally:
        ops.set_matmul_variant(rt, -1)
    return {k2: {"min": min(v), "median": round(statistics.median(v), 1), "max": max(v)} for k2, v in res.items()}


def clock(rt, w, m, n, k, ta, tb, zeros=False):
    dt = torch.bfloat16
    a = torch.randn((k, m) if ta else (m, k), device="cuda").to(dt)
    b = torch.randn((n, k) if tb else (k, n), device="cuda").to(dt)
    if zeros:  # the same instruction stream on operands that toggle nothing: what the DATA costs in clock
        a.zero_()
        b.zero_()
    out = torch.empty(m, n, device="cuda", dtype=dt)
    ops.set_matmul_variant(rt, w)
    try:
        for _ in range(30):
            ops.matmul(rt, a, b, trans_a=ta, trans_b=tb, out=out)
        rt.sync()
    finally:
        ops.set_matmul_variant(rt, -1)
    tc, tr = out.view(torch.int64).flatten()[:2].tolist()
    us = tr / 100.0
    mhz = tc / max(tr, 1) * 100
    tf = 2.0 * m * n * k / max(us, 1e-9) / 1e6
    return {"shape": [m, n, k], "ta": ta, "tb": tb, "operands": "zeros" if zeros else "N(0,1)", "core_MHz": round(mhz, 1), "K_loops_us_workgroup0": us, "TFLOPs_workgroup0": round(tf, 1),
            "issue_efficiency": round(tf / (2500.0 * mhz / 2400.0), 3)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--ab", action="store_true")
    ap.add_argument("--clock", action="store_true")
    a = ap.parse_args()
    rt = RocmRuntime(0)
    w = ops.matmul_variants().index("wave128")
    if a.clock:
        import os

        if os.environ.get("IROCM_W128_DBG") != "8":
            sys.exit("run with IROCM_W128_DBG=8")
        for rep in range(3):
            for (m, n, k, ta, tb) in [(4096, 4096, 4096, False, False), (4096, 4096, 4096, True, False), (8192, 8192, 8192, False, False)]:
                print(json.dumps(clock(rt, w, m, n, k, ta, tb)), flush=True)
            print(json.dumps(clock(rt, w, 4096, 4096, 4096, False, False, zeros=True)), flush=True)
        sys.exit(0)
    if a.ab:
        for dt in (torch.bfloat16, torch.float16):
            for (m, n, k, ta, tb) in [(4096, 4096, 4096, False, False), (4096, 4096, 4096, False, True), (4096, 4096, 4096, True, False), (4096, 4096, 4096, True, True),
                                      (8192, 4096, 4096, False, False), (4096, 8192, 2048, False, False), (8192, 8192, 8192, False, False), (16384, 4096, 1024, False, False),
                                      (16384, 3072, 768, False, False), (16384, 768, 3072, False, False), (2048, 4096, 4096, False, False)]:
                print(json.dumps({"dtype": str(dt).split(".")[-1], "shape": [m, n, k], "ta": ta, "tb": tb, **ab(rt, w, dt, m, n, k, ta, tb)}), flush=True)
        sys.exit(0)
    if not a.no_check:
        fails = 0
        for dt in (torch.bfloat16, torch.float16):
            for (b, m, n, k) in [(1, 256, 256, 128), (1, 256, 256, 512), (1, 512, 768, 256), (3, 512, 256, 384), (1, 1280, 2304, 640), (1, 4096, 4096, 1024)]:
                for ta in (False, True):
                    for tb in (False, True):
                        r = check(rt, dt, b, m, n, k, ta, tb)
                        if r["bad"] or r["variant"] != "wave128":
                            fails += 1
                            print("FAIL", json.dumps(r), flush=True)
        print(f"parity: {fails} failing cases", flush=True)
    for dt in (torch.bfloat16, torch.float16):
        for (m, n, k) in [(4096, 4096, 4096), (8192, 8192, 8192), (16384, 768, 3072), (16384, 3072, 768), (16384, 2304, 768), (2048, 11008, 4096), (2048, 4096, 11008)]:
            for tb in (False, True):
                row = {"dtype": str(dt).split(".")[-1], "shape": [m, n, k], "trans_b": tb}
                for rep in range(2):
                    row[f"shipped_rep{rep}"], row["shipped_variant"] = timeit(rt, dt, m, n, k, tb, 4, a.reps)
                    row[f"wave128_rep{rep}"], _ = timeit(rt, dt, m, n, k, tb, w, a.reps)
                print(json.dumps(row), flush=True)
