"""Random transformer-style blocks IN THE FORM pyinfinitensor/onnx.py EMITS THEM, through the reference executor + ROCM
plugin, launch planning ON vs OFF (f16): linear layers as MatMul -> Add(bias) with the bias on either side, the q / k / v
projections issued and reshaped in random interleavings, K^T as Transpose(0,1,3,2) / merged Transpose(0,2,3,1) / transB, the
attention chain with or without scale and mask, the head merge, LayerNorm and Gelu as single operators or decomposed into
their opset < 17 primitives, residual joins — with only the final tensor kept alive, so the memory planner recycles every
buffer it can. Fused kernels round once where the chain rounds per operator, so outputs are compared within a 16-bit
tolerance AND against the fp64 oracle of the same graph; the point is to catch a planned launch that reads a buffer after
someone recycled it, writes one early, or absorbs an operator it should not have.
python tools/onnx_form_fuzz.py [n_graphs]   (FUZZ_SEED in the environment; FUZZ_ONLY=<g> runs one graph with its plan)"""
import os
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
from conftest import load_backend_module  # noqa: E402

B = load_backend_module()
assert B is not None and hasattr(B, "RocmRuntime"), "plugin build missing"
rocm = B.RocmRuntime(0)
F16 = 10
n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(os.environ.get("FUZZ_SEED", "31"))
lin = B.ActType.Linear
f64 = lambda a: np.asarray(a).astype(np.float64)
r16 = lambda a: np.asarray(a, np.float64).astype(np.float16).astype(np.float64)


def gelu64(x):
    from scipy.special import erf

    return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))


def build(seed, h, feeds):
    """Returns (output tensor, fp64 value of it with every operator result rounded to f16 like the unfused graph)."""
    rng = np.random.default_rng(seed)
    Bt, S = int(rng.choice([1, 2])), int(rng.choice([64, 128, 256]))
    NH, D = int(rng.choice([1, 2, 4])), int(rng.choice([64, 128]))
    H = NH * D
    F = int(rng.choice([2, 4])) * H
    decomposed = rng.random() < 0.5

    def weight(a):
        t = h.tensor(list(a.shape), F16)
        t.set_weight()
        a = a.astype(np.float16)
        feeds.append((t, a))
        return t, f64(a)

    consts = {}
    if decomposed:
        for name, v in (("two", 2.0), ("one", 1.0), ("half", 0.5), ("sqrt2", np.sqrt(2.0)), ("eps", 1e-5)):
            consts[name] = weight(np.array([v]))

    def linear(x, xv, cin, cout, scale=None):
        w, wv = weight(rng.standard_normal((cin, cout)) * (scale or 1.0 / np.sqrt(cin)))
        b, bv = weight(rng.standard_normal((cout,)) * 0.2)
        mm = h.matmul(x, w, None, False, False, None, lin, "default")
        mv = r16(xv @ wv)
        y = h.add(b, mm, None) if rng.random() < 0.5 else h.add(mm, b, None)
        return y, r16(mv + bv)

    def ln(x, xv):
        g, gv = weight(1 + 0.1 * rng.standard_normal(H))
        b, bv = weight(0.1 * rng.standard_normal(H))
        if not decomposed:
            y = h.layerNormalization(x, g, None, b, 1e-5, 2, 1)
            mu = xv.mean(-1, keepdims=True)
            var = ((xv - mu) ** 2).mean(-1, keepdims=True)
            return y, r16((xv - mu) / np.sqrt(var + 1e-5) * gv + bv)
        two, eps = consts["two"][0], consts["eps"][0]
        epsv = consts["eps"][1][0]
        d = h.sub(x, h.reduceMean(x, None, [2], True), None)
        var = h.reduceMean(h.pow(d, two, None), None, [2], True)
        y = h.add(h.mul(h.div(d, h.sqrt(h.add(var, eps, None), None), None), g, None), b, None)
        mu = r16(xv.mean(-1, keepdims=True))
        dv = r16(xv - mu)
        vv = r16(r16(dv * dv).mean(-1, keepdims=True))
        return y, r16(r16(r16(dv / r16(np.sqrt(r16(vv + epsv)))) * gv) + bv)

    def gelu(x, xv):
        if not decomposed:
            return h.gelu(x, None), r16(gelu64(xv))
        from scipy.special import erf

        sqrt2, one, half = consts["sqrt2"][0], consts["one"][0], consts["half"][0]
        e = h.add(h.erf(h.div(x, sqrt2, None), None), one, None)
        ev = r16(r16(erf(r16(xv / consts["sqrt2"][1][0]))) + 1.0)
        if rng.random() < 0.5:
            return h.mul(h.mul(x, e, None), half, None), r16(r16(xv * ev) * 0.5)
        return h.mul(h.mul(x, half, None), e, None), r16(r16(xv * 0.5) * ev)

    x = h.tensor([Bt, S, H], F16)
    xa = rng.standard_normal((Bt, S, H)).astype(np.float16)
    feeds.append((x, xa))
    xv = f64(xa)
    cur, curv = h.relu(x, None), np.maximum(xv, 0)
    for _ in range(int(rng.integers(1, 3))):
        kind = int(rng.integers(0, 3))
        if kind == 0:  # attention block
            hd = lambda t: h.transpose(h.reshape(t, None, [Bt, S, NH, D]), None, [0, 2, 1, 3])
            hdv = lambda v: v.reshape(Bt, S, NH, D).transpose(0, 2, 1, 3)
            kt_form = int(rng.integers(0, 3))  # 0: Transpose(0,1,3,2) of the split K, 1: merged (0,2,3,1), 2: transB
            order = int(rng.integers(0, 3))
            lq, lqv = linear(cur, curv, H, H)
            lk, lkv = linear(cur, curv, H, H)
            lv, lvv = linear(cur, curv, H, H)
            qv_, kv_, vv_ = hdv(lqv), hdv(lkv), hdv(lvv)
            if order == 0:  # the exporter's: k, v split first, q last
                if kt_form == 1:
                    kx = h.transpose(h.reshape(lk, None, [Bt, S, NH, D]), None, [0, 2, 3, 1])
                else:
                    k = hd(lk)
                v = hd(lv)
                q = hd(lq)
            else:
                q = hd(lq)
                if kt_form == 1:
                    kx = h.transpose(h.reshape(lk, None, [Bt, S, NH, D]), None, [0, 2, 3, 1])
                else:
                    k = hd(lk)
                v = hd(lv)
            if kt_form == 0:
                kx = h.transpose(k, None, [0, 1, 3, 2])
            if kt_form == 2:
                s = h.matmul(q, k, None, False, True, None, lin, "default")
            else:
                s = h.matmul(q, kx, None, False, False, None, lin, "default")
            sv = r16(qv_ @ kv_.transpose(0, 1, 3, 2))
            if rng.random() < 0.8:
                sc, scv = weight(np.array([np.sqrt(D)]))
                s, sv = h.div(s, sc, None), r16(sv / scv[0])
            if rng.random() < 0.6:
                m = np.where(rng.random((Bt, 1, 1, S)) < 0.85, 0.0, -10000.0)
                mt, mv = weight(m)
                s, sv = h.add(s, mt, None), r16(sv + mv)
            p = h.softmax(s, None, 3)
            e = np.exp(sv - sv.max(-1, keepdims=True))
            pv = r16(e / e.sum(-1, keepdims=True))
            ctx = h.matmul(p, v, None, False, False, None, lin, "default")
            cv = r16(pv @ vv_)
            ctx = h.reshape(h.transpose(ctx, None, [0, 2, 1, 3]), None, [Bt, S, H])
            cv = cv.transpose(0, 2, 1, 3).reshape(Bt, S, H)
            o, ov = linear(ctx, cv, H, H)
            cur, curv = ln(h.add(o, cur, None), r16(ov + curv))
        elif kind == 1:  # feed-forward block
            u, uv = linear(cur, curv, H, F)
            a, av = gelu(u, uv)
            dn, dv = linear(a, av, F, H)
            cur, curv = ln(h.add(dn, cur, None), r16(dv + curv))
        else:  # a bare linear + norm
            y, yv = linear(cur, curv, H, H)
            cur, curv = ln(y, yv)
    return h.abs(cur, None), np.abs(curv)


only = os.environ.get("FUZZ_ONLY")
bad = 0
TOTAL = 0
for g in range(n_graphs):
    if only is not None and g != int(only):
        continue
    seed = seed0 * 1000 + g
    got, counts = {}, {}
    want = None
    try:
        for on in (True, False):
            rocm.set_fusion(on)
            h = B.GraphHandler(rocm)
            feeds = []
            out, want = build(seed, h, feeds)
            h.data_malloc()
            for t, a in feeds:
                t.copyin_numpy(np.ascontiguousarray(a))
            if only is not None and on:
                print("\n".join(h.rocm_fusion_plan()))
            c0 = rocm.fused_launch_count()
            if g % 2:
                h.run_with_hipgraph()
            else:
                h.run()
            counts[on] = rocm.fused_launch_count() - c0
            got[on] = out.copyout_numpy().astype(np.float64).reshape(want.shape)
    finally:
        rocm.set_fusion(True)
    TOTAL += counts[True]
    scale = max(1e-6, float(np.abs(want).max()))
    e_onoff = float(np.abs(got[True] - got[False]).max()) / scale
    e_on = float(np.abs(got[True] - want).max()) / scale
    e_off = float(np.abs(got[False] - want).max()) / scale
    # the unfused graph rounds exactly like the oracle restatement (up to kernel-internal orderings); the fused one rounds less
    # (a graph may legitimately plan nothing: a lone linear whose bias Add's output the planner put on the MatMul's dead input)
    if not (np.isfinite(got[True]).all() and e_onoff <= 2e-2 and e_on <= 2e-2 and e_off <= 2e-2 and counts[False] == 0):
        bad += 1
        print(f"FAIL graph seed {seed}: on-vs-off {e_onoff:.3g}, on-vs-oracle {e_on:.3g}, off-vs-oracle {e_off:.3g} of scale {scale:.3g}; "
              f"fused launches {counts}", flush=True)
    elif g < 6 or g % 10 == 0 or only is not None:
        print(f"ok graph seed {seed}: fused launches {counts[True]}, on-vs-off {e_onoff:.2g}, on-vs-oracle {e_on:.2g}", flush=True)
print(f"{n_graphs - bad}/{n_graphs} graphs agree with planning on / off and with the oracle ({TOTAL} fused launches)")
sys.exit(1 if bad else 0)
