"""Randomised sweeps of the row / element-wise / movement kernels against torch (CPU, fp32 or exact): softmax over any axis,
LayerNorm over trailing axes, RMSNorm, ReduceSum / ReduceMean over random axis sets, broadcast binaries, Transpose with
random permutations, Concat / Split, Pad. python tools/rowops_fuzz.py [n] (FUZZ_SEED in the environment)."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from infinitensor_amd import RocmRuntime, ops

rt = RocmRuntime(0)
rt.use_torch_stream()
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "11")))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
DT = [torch.float32, torch.float16, torch.bfloat16]
TOL = {torch.float32: 2e-5, torch.float16: 3e-3, torch.bfloat16: 2e-2}
bad = 0


def shape(maxrank=4, maxdim=70, maxel=400000):
    while True:
        s = [int(rng.integers(1, maxdim)) for _ in range(int(rng.integers(1, maxrank + 1)))]
        if int(np.prod(s)) <= maxel:
            return s


def close(y, ref, dt, what, exact=False):
    global bad
    y = y.float().cpu()
    ok = tuple(y.shape) == tuple(ref.shape) and (torch.equal(y, ref.float()) if exact else
                                                 bool(((y - ref.float()).abs() <= TOL[dt] * (1 + ref.float().abs())).all()))
    if not ok:
        bad += 1
        print("FAIL", what, flush=True)


for case in range(n_cases):
    dt = DT[case % 3]
    kind = case % 9
    s = shape()
    x = torch.randn(s).to(dt)
    xd = x.cuda()
    try:
        if kind == 0:
            ax = int(rng.integers(-len(s), len(s)))
            close(ops.softmax(rt, xd, ax), torch.softmax(x.float(), ax), dt, f"softmax {s} axis {ax} {dt}")
        elif kind == 1:
            ax = int(rng.integers(0, len(s)))
            ns = s[ax:]
            g, b = torch.randn(ns).to(dt), torch.randn(ns).to(dt)
            ref = torch.nn.functional.layer_norm(x.float(), ns, g.float(), b.float(), 1e-5)
            close(ops.layer_norm(rt, xd, g.cuda(), b.cuda(), 1e-5, ax), ref, dt, f"layer_norm {s} axis {ax} {dt}")
        elif kind == 2:
            g = torch.randn(s[-1]).to(dt)
            ref = x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-5) * g.float()
            close(ops.rms_norm(rt, xd, g.cuda(), 1e-5), ref, dt, f"rms_norm {s} {dt}")
        elif kind == 3:
            axes = sorted(set(int(a) for a in rng.integers(0, len(s), size=int(rng.integers(1, len(s) + 1)))))
            keep = bool(rng.integers(0, 2))
            which = ["sum", "mean"][int(rng.integers(0, 2))]
            ref = getattr(x.float(), which)(dim=axes, keepdim=keep)
            y = ops.reduce(rt, which, xd, axes, keep)
            n_red = int(np.prod([s[a] for a in axes]))
            scale = 1.0 if which == "mean" else max(1.0, n_red ** 0.5)
            ok = tuple(y.shape) == tuple(ref.shape) and bool(((y.float().cpu() - ref).abs() <= TOL[dt] * scale * (1 + ref.abs())).all())
            if not ok:
                bad += 1
                print("FAIL", f"reduce {which} {s} axes {axes} keep {keep} {dt}", flush=True)
        elif kind == 4:
            s2 = [d if rng.random() < 0.6 else 1 for d in s][int(rng.integers(0, len(s))):]
            y = (torch.randn(s2).abs() + 0.5).to(dt)
            op = ["add", "sub", "mul", "div", "max", "min"][int(rng.integers(0, 6))]
            f = {"add": torch.add, "sub": torch.sub, "mul": torch.mul, "div": torch.div, "max": torch.maximum, "min": torch.minimum}[op]
            # the offset operand (|value| mostly > 0.5) is the divisor: a quotient beyond the 16-bit range says nothing about the kernel
            a, b = (xd, y.cuda()) if (op == "div" or rng.random() < 0.5) else (y.cuda(), xd)
            ref = f(a.float().cpu(), b.float().cpu())
            close(ops.binary(rt, op, a, b), ref, dt, f"binary {op} {list(a.shape)} {list(b.shape)} {dt}")
        elif kind == 5:
            perm = [int(p) for p in rng.permutation(len(s))]
            close(ops.transpose(rt, xd, perm), x.permute(perm).contiguous(), dt, f"transpose {s} {perm} {dt}", exact=True)
        elif kind == 6:
            ax = int(rng.integers(0, len(s)))
            parts = []
            for _ in range(int(rng.integers(1, 4))):
                sp = list(s)
                sp[ax] = int(rng.integers(1, 9))
                parts.append(torch.randn(sp).to(dt))
            cat = torch.cat(parts, ax)
            close(ops.concat(rt, [p.cuda() for p in parts], ax), cat, dt, f"concat {s} axis {ax} {dt}", exact=True)
            back = ops.split(rt, cat.cuda(), ax, [p.shape[ax] for p in parts])
            for p_, q_ in zip(parts, back):
                close(q_, p_, dt, f"split {list(cat.shape)} axis {ax} {dt}", exact=True)
        elif kind == 7:
            pads = [int(v) for v in rng.integers(0, 3, size=2 * len(s))]
            tp = []
            for d in range(len(s) - 1, -1, -1):
                tp += [pads[d], pads[len(s) + d]]
            close(ops.pad(rt, xd, pads), torch.nn.functional.pad(x.float(), tp).to(dt), dt, f"pad {s} {pads} {dt}", exact=True)
        else:
            op = ["relu", "sigmoid", "tanh", "abs", "neg", "exp", "gelu", "silu"][int(rng.integers(0, 8))]
            f = {"relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh, "abs": torch.abs, "neg": torch.neg, "exp": torch.exp,
                 "gelu": torch.nn.functional.gelu, "silu": torch.nn.functional.silu}[op]
            close(ops.unary(rt, op, xd), f(x.float()), dt, f"unary {op} {s} {dt}")
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL (exception)", kind, s, dt, repr(e)[:160], flush=True)
print(f"{n_cases - bad}/{n_cases} cases ok")
sys.exit(1 if bad else 0)
