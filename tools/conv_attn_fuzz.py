"""Randomised sweeps of infini_rocm_conv2d_res and infini_rocm_attention_ex against torch fp32 references on the same rounded
inputs (CPU): kernel sizes, strides, pads, groups, ragged planes, bias / residual / relu; attention head sizes, ragged
sequence lengths, key masks, full masks, causal, scale sign. python tools/conv_attn_fuzz.py [n_conv] [n_attn]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import torch.nn.functional as F

from infinitensor_amd import RocmRuntime, ops

rt = RocmRuntime(0)
rt.use_torch_stream()
import os
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "77")))
n_conv = int(sys.argv[1]) if len(sys.argv) > 1 else 120
n_attn = int(sys.argv[2]) if len(sys.argv) > 2 else 80
bad = 0
for case in range(n_conv):
    dt = [torch.float16, torch.bfloat16][case % 2]
    groups = int(rng.choice([1, 1, 1, 2, 4]))
    c = groups * int(rng.choice([1, 3, 8, 16, 32, 64, 128]))
    f = groups * int(rng.choice([1, 4, 16, 32, 64, 96, 128, 200]))
    r = int(rng.choice([1, 1, 3, 3, 5, 7]))
    s = r if rng.random() < 0.8 else int(rng.choice([1, 3]))
    sh = int(rng.choice([1, 1, 2]))
    sw = sh
    dh = int(rng.choice([1, 1, 2])) if sh == 1 else 1
    h = int(rng.integers(max(1, (r - 1) * dh + 1), 40))
    w = int(rng.integers(max(1, (s - 1) * dh + 1), 40))
    ph = int(rng.choice([0, (r - 1) * dh // 2, 1]))
    pw = int(rng.choice([0, (s - 1) * dh // 2, 1]))
    n = int(rng.choice([1, 2, 5]))
    if c * f * r * s * h * w * n > 3e8:
        continue
    x = torch.randn(n, c, h, w).to(dt)
    wt = (torch.randn(f, c // groups, r, s) / max(1.0, (c // groups * r * s) ** 0.5) * 2).to(dt)
    bias = torch.randn(f).to(dt) if rng.random() < 0.6 else None
    act = int(rng.choice([0, 1, 1]))
    ref = F.conv2d(x.float(), wt.float(), None if bias is None else bias.float(), (sh, sw), (ph, pw), (dh, dh), groups)
    if ref.numel() == 0:
        continue
    res = torch.randn(ref.shape).to(dt) if rng.random() < 0.4 else None
    if res is not None:
        ref = ref + res.float()
    if act == 1:
        ref = torch.relu(ref)
    variant = int(rng.choice([-1, -1, -1, 1, 2, 4]))
    ops.set_conv_variant(rt, variant) if hasattr(ops, "set_conv_variant") else None
    try:
        y = ops.conv2d(rt, x.cuda(), wt.cuda(), ph, pw, sh, sw, dh, dh, None if bias is None else bias.cuda(), act,
                       residual=None if res is None else res.cuda())
        torch.cuda.synchronize()
        ok = tuple(y.shape) == tuple(ref.shape)
        tol = 5e-3 if dt == torch.float16 else 3e-2
        err = ((y.float().cpu() - ref).abs() / (ref.abs() + 1)).max().item() if ok else "shape"
        ok = ok and err <= tol
    except Exception as e:  # noqa: BLE001
        ok, err = False, repr(e)[:120]
    finally:
        ops.set_conv_variant(rt, -1) if hasattr(ops, "set_conv_variant") else None
    if not ok:
        bad += 1
        print(f"FAIL conv {case}: {dt} n{n} c{c} {h}x{w} f{f} {r}x{s} s{sh} d{dh} p{ph},{pw} g{groups} bias{int(bias is not None)} res{int(res is not None)} act{act} v{variant}: {err}", flush=True)
print(f"conv: {bad} failures")
abad = 0
for case in range(n_attn):
    dt = [torch.float16, torch.bfloat16][case % 2]
    d = int(rng.choice([64, 128]))
    b, hh = int(rng.choice([1, 2, 3])), int(rng.choice([1, 2, 5]))
    sq, sk = int(rng.integers(1, 400)), int(rng.integers(1, 400))
    causal = bool(rng.random() < 0.4)
    if causal and sk < sq:
        sk = sq
    kind = int(rng.choice([0, 1, 1, 2]))  # none, key mask [b, sk], full [b, sq, sk]
    scale = float(rng.choice([d ** -0.5, d ** -0.5, 0.3, -0.2]))
    q, k, v = (torch.randn(b, hh, s_, d).to(dt) for s_ in (sq, sk, sk))
    mask = None
    s_ref = torch.matmul(q.float(), k.float().transpose(-1, -2)) * scale
    if kind == 1:
        mask = torch.where(torch.rand(b, sk) < 0.8, 0.0, -10000.0).to(dt)
        mask[:, 0] = 0
        s_ref = s_ref + mask.float()[:, None, None, :]
    elif kind == 2:
        mask = (torch.randn(b, sq, sk) * 2).to(dt)
        s_ref = s_ref + mask.float()[:, None, :, :]
    if causal:
        keep = torch.tril(torch.ones(sq, sk, dtype=torch.bool), diagonal=sk - sq)
        s_ref = s_ref.masked_fill(~keep, float("-inf"))
    ref = torch.matmul(torch.softmax(s_ref, -1), v.float())
    try:
        y = ops.attention(rt, q.cuda(), k.cuda(), v.cuda(), scale, None if mask is None else mask.cuda(), causal)
        torch.cuda.synchronize()
        tol = 4e-3 if dt == torch.float16 else 2.5e-2
        err = (y.float().cpu() - ref).abs().max().item()
        ok = err <= tol
    except Exception as e:  # noqa: BLE001
        ok, err = False, repr(e)[:120]
    if not ok:
        abad += 1
        print(f"FAIL attn {case}: {dt} b{b} h{hh} sq{sq} sk{sk} d{d} causal{int(causal)} mask{kind} scale{scale:.3f}: {err}", flush=True)
print(f"attention: {abad} failures")
sys.exit(1 if bad or abad else 0)
