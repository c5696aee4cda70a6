"""Random operator graphs through the reference executor + ROCM plugin, launch-time fusion ON vs OFF: the outputs must be
bit-identical (every fusion rule claims that for f16: head split, grouped / hoisted / parked MatMuls, Silu -> Mul, RoPE head
split, copy elision). Graphs mix MatMuls that share activations with unary / binary element-wise ops, RoPE and the
Reshape -> Transpose head split, in random order, with only the final tensors kept alive (so the planner recycles buffers).
INFINI_ROCM_FUSE_GELU=0 python tools/fusion_fuzz.py [n_graphs]   (FUZZ_SEED in the environment; the MatMul -> Gelu epilogue
rounds once instead of twice and is NOT bit-identical by design: with it on, Gelu outputs differ in the last place)"""
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
F16, U32 = 10, 12
n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(os.environ.get("FUZZ_SEED", "101"))
lin = B.ActType.Linear
bad = 0
TOTAL = [0, 0]


TRACE = []


def build(seed, h, feeds):
    """Returns the list of output tensors; appends (tensor, array) to feeds."""
    rng = np.random.default_rng(seed)
    TRACE.clear()
    Bt, S = int(rng.choice([1, 2, 4])), int(rng.choice([128, 256, 512]))
    NH, D = int(rng.choice([1, 2, 4])), 128
    if rng.random() < 0.35:  # big enough that a two-member group cannot run split-K: the parked-member path
        Bt, S, NH = 8, 512, 12
    H = NH * D
    T = Bt * S

    def weight(shape, scale):
        t = h.tensor(list(shape), F16)
        t.set_weight()
        feeds.append((t, (rng.standard_normal(shape) * scale).astype(np.float16)))
        return t

    x = h.tensor([Bt, S, H], F16)
    feeds.append((x, rng.standard_normal((Bt, S, H)).astype(np.float16)))
    pos = h.tensor([Bt, S], U32)
    pos.set_weight()
    feeds.append((pos, np.tile(np.arange(S, dtype=np.uint32), (Bt, 1))))
    live = [h.relu(x, None)]  # same-shape [Bt, S, H] activations
    outs = []
    un = [h.relu, h.silu, h.neg, h.abs, h.sigmoid, h.tanh]
    for _ in range(int(rng.integers(4, 12))):
        k = int(rng.integers(0, 9))
        ai = int(rng.integers(0, len(live)))
        a = live[ai]
        TRACE.append(f"k{k}(a=live[{ai}]/{len(live)})")
        if k <= 2:  # one to three MatMuls of the same activation, consumers in random order
            n_mm = int(rng.integers(1, 4))
            prods = [h.matmul(a, weight((H, H), 0.06), None, False, False, None, lin, "default") for _ in range(n_mm)]
            order = rng.permutation(n_mm)
            res = []
            for j in order:
                c = int(rng.integers(0, 5))
                p = prods[j]
                TRACE.append(f"  mm{j}->c{c}")
                if c == 0:
                    res.append(un[int(rng.integers(0, len(un)))](p, None))
                elif c == 1 and len(res):
                    res.append(h.mul(res[-1], p, None))
                elif c == 2:  # head split
                    outs.append(h.transpose(h.reshape(p, None, [Bt, S, NH, D]), None, [0, 2, 1, 3]))
                elif c == 3:  # RoPE, sometimes with the head split behind it
                    r = h.RoPE(pos, p, None)
                    if rng.random() < 0.6:
                        outs.append(h.transpose(h.reshape(r, None, [Bt, S, NH, D]), None, [0, 2, 1, 3]))
                    else:
                        res.append(r)
                else:
                    res.append(p)
            live += res
        elif k == 3:
            live.append(un[int(rng.integers(0, len(un)))](a, None))
        elif k == 4:
            b = live[int(rng.integers(0, len(live)))]
            live.append([h.add, h.mul, h.sub][int(rng.integers(0, 3))](a, b, None))
        elif k == 5:  # gated pair
            g = h.matmul(a, weight((H, H), 0.06), None, False, False, None, lin, "default")
            u = h.matmul(a, weight((H, H), 0.06), None, False, False, None, lin, "default")
            live.append(h.mul(h.silu(g, None), u, None))
        elif k == 6:
            live.append(h.gelu(h.matmul(a, weight((H, H), 0.06), None, False, False, weight((H,), 0.5), lin, "default"), None))
        elif k == 7:  # the gated pair in a decoder's operator order: mm_g, Silu, mm_u, Mul (the planner recycles mm_g's buffer)
            sg = h.silu(h.matmul(a, weight((H, H), 0.06), None, False, False, None, lin, "default"), None)
            live.append(h.mul(sg, h.matmul(a, weight((H, H), 0.06), None, False, False, None, lin, "default"), None))
        else:  # q / k in a decoder's order: mm_q, RoPE, [split], mm_k, RoPE, [split]
            for _q in range(2):
                r = h.RoPE(pos, h.matmul(a, weight((H, H), 0.06), None, False, False, None, lin, "default"), None)
                if rng.random() < 0.7:
                    outs.append(h.transpose(h.reshape(r, None, [Bt, S, NH, D]), None, [0, 2, 1, 3]))
                else:
                    live.append(r)
        if len(live) > 4:  # let tensors die so that buffers get recycled
            live = live[-4:]
    # a tensor that has consumers is not a graph output: the planner recycles its buffer after the last one. Compare only
    # tensors nobody reads: the head-split results and a fresh copy of the last activation
    return outs + [h.abs(live[-1], None)]


only = os.environ.get("FUZZ_ONLY")
for g in range(n_graphs):
    if only is not None and g != int(only):
        continue
    seed = seed0 * 1000 + g
    got = {}
    counts = {}
    try:
        for on in (True, False):
            rocm.set_fusion(on)
            h = B.GraphHandler(rocm)
            feeds = []
            outs = build(seed, h, feeds)
            h.data_malloc()
            for t, a in feeds:
                t.copyin_numpy(np.ascontiguousarray(a))
            c0 = (rocm.fused_launch_count(), rocm.parked_member_count())
            if g % 2:
                h.run_with_hipgraph()
            else:
                h.run()
            counts[on] = (rocm.fused_launch_count() - c0[0], rocm.parked_member_count() - c0[1])
            got[on] = [o.copyout_numpy() for o in outs]
    finally:
        rocm.set_fusion(True)
    TOTAL[0] += counts[True][0]
    TOTAL[1] += counts[True][1]
    ok = all(np.array_equal(a.view(np.uint16), b.view(np.uint16)) for a, b in zip(got[True], got[False]))
    fin = all(np.isfinite(a.astype(np.float32)).all() for a in got[False])
    if not ok:
        bad += 1
        worst = max(float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) for a, b in zip(got[True], got[False]))
        print(f"FAIL graph seed {seed}: fused launches {counts[True][0]}, parked {counts[True][1]}, max abs diff {worst}; per output: "
              + ", ".join(f"{a.shape}:{float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()):.3g}" for a, b in zip(got[True], got[False])), flush=True)
        print("   trace:", " ".join(TRACE), flush=True)
        for a, b in zip(got[True], got[False]):
            d = np.flatnonzero(a.view(np.uint16).ravel() != b.view(np.uint16).ravel())
            if d.size:
                print(f"   output {a.shape}: {d.size} of {a.size} elements differ, flat index {d[0]} .. {d[-1]}", flush=True)
    elif g < 8 or g % 10 == 0 or only is not None:
        print(f"ok graph seed {seed}: {len(got[True])} outputs, fused launches {counts[True][0]}, parked {counts[True][1]}, finite {fin}", flush=True)
print(f"{n_graphs - bad}/{n_graphs} graphs bit-identical with fusion on / off "
      f"({TOTAL[0]} fused launches, {TOTAL[1]} parked group members in the fused runs)")
sys.exit(1 if bad else 0)
