"""Randomised shape sweep of infini_rocm_matmul against torch's fp32 matmul (same rounded inputs): layouts, batch broadcast,
bias forms, activations, every kernel variant. python tools/gemm_fuzz.py [n_cases] — prints the failures, exits non-zero on any."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from infinitensor_amd import RocmRuntime, ops

rt = RocmRuntime(0)
rt.use_torch_stream()
import os
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "2024")))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
ACT = {0: lambda x: x, 1: torch.relu, 2: torch.sigmoid, 3: torch.tanh, 4: lambda x: torch.nn.functional.gelu(x),
       5: lambda x: torch.nn.functional.gelu(x)}
W128 = ops.matmul_variants().index("wave128")
bad = 0
for case in range(n_cases):
    dt = [torch.float16, torch.bfloat16][case % 2]
    big = rng.random() < 0.5
    m = int(rng.choice([1, 7, 64, 200, 256, 257, 512, 1000, 1024, 2048, 3000])) if big else int(rng.integers(1, 300))
    n = int(rng.choice([8, 64, 192, 256, 264, 768, 1000, 1024, 1536, 2304])) if big else int(rng.integers(1, 300))
    k = int(rng.choice([64, 128, 192, 256, 320, 768, 1024])) if big else int(rng.integers(1, 200))
    batch = int(rng.choice([1, 1, 1, 2, 3]))
    ta, tb = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    act = int(rng.choice([0, 0, 1, 5, 2, 3, 4]))
    variant = int(rng.choice([-1, -1, 0, 1, 2, 3, 4, 5, 6]))
    bias_kind = int(rng.choice([0, 1, 1, 2, 3]))  # none, row [n], column [m,1], full [m,n]
    w128 = case % 5 == 4  # every fifth case: the four-wave kernel's contract (whole 256^2 tiles, K % 128 == 0, plain store), forced
    if w128:
        m, n, k = 256 * int(rng.integers(1, 14)), 256 * int(rng.integers(1, 10)), 128 * int(rng.integers(1, 34))
        act, bias_kind, variant = 0, 0, W128
    a_shape = (batch, k, m) if ta else (batch, m, k)
    bcast_b = batch > 1 and rng.random() < 0.5
    b_shape = ((k, n) if not tb else (n, k)) if bcast_b else ((batch, k, n) if not tb else (batch, n, k))
    a = torch.randn(a_shape, device="cuda").to(dt)
    b = (torch.randn(b_shape, device="cuda") / max(1.0, k ** 0.5) * 2).to(dt)
    bias = None
    if bias_kind == 1:
        bias = torch.randn(n, device="cuda").to(dt)
    elif bias_kind == 2:
        bias = torch.randn(m, 1, device="cuda").to(dt)
    elif bias_kind == 3:
        bias = torch.randn(m, n, device="cuda").to(dt)
    af = a.float().transpose(-1, -2) if ta else a.float()
    bf = b.float().transpose(-1, -2) if tb else b.float()
    ref = af @ bf
    if bias is not None:
        ref = ref + bias.float()
    ref = ACT[act](ref)
    ops.set_matmul_variant(rt, variant)
    try:
        y = ops.matmul(rt, a, b, bias, ta, tb, act=act)
        torch.cuda.synchronize()
        if w128 and ops.matmul_last_variant(rt) != "wave128":
            raise RuntimeError("wave128 refused a problem inside its contract")
        tol = 4e-3 if dt == torch.float16 else 2.5e-2
        err = ((y.float() - ref).abs() / (ref.abs() + 1)).max().item()
        ok = err <= tol and bool(torch.isfinite(y.float()).all())
    except Exception as e:  # noqa: BLE001
        ok, err = False, repr(e)[:120]
    finally:
        ops.set_matmul_variant(rt, -1)
    if not ok:
        bad += 1
        print(f"FAIL case {case}: {dt} m{m} n{n} k{k} b{batch} ta{int(ta)} tb{int(tb)} bcastB{int(bcast_b)} bias{bias_kind} act{act} v{variant}: {err}", flush=True)
print(f"{n_cases - bad}/{n_cases} cases ok")
sys.exit(1 if bad else 0)
