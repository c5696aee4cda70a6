"""The launch planner (plugin/src/rocm_fusion.cc) WITHOUT a GPU: `GraphHandler.rocm_fusion_plan()` plans a graph that lives on
the reference's native-CPU runtime (nothing is launched, device-address alignment is assumed) and reports the items. Checked
here: the chains the ONNX front-end really emits (pyinfinitensor/onnx.py — conv -> reshape(bias) -> add, MatMul -> Add(bias),
Transpose(K) -> MatMul, opset < 17 LayerNorm / Gelu, the exporter's operator order) are recognised as ONE launch each."""
import sys
from collections import Counter
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
F32, F16 = 1, 10


@pytest.fixture(scope="module")
def B(plugin_backend):
    if not hasattr(plugin_backend.GraphHandler, "rocm_fusion_plan"):
        pytest.skip("plugin build predates the planner")
    return plugin_backend


def plan_of(h):
    """[(slot, what, [members])]; what == 'op' for an operator that runs its own kernel."""
    out = []
    for line in h.rocm_fusion_plan():
        slot, rest = line.split(" ", 1)
        what, mem = rest.rsplit(" [", 1)
        out.append((int(slot), what, [int(m) for m in mem.rstrip("]").split(",")]))
    return out


def weight(h, arr, code=F16):
    t = h.tensor(list(arr.shape), code)
    t.set_weight()
    return t, np.ascontiguousarray(arr.astype(np.float16 if code == F16 else np.float32))


def finish(h, feeds):
    h.data_malloc()
    for t, a in feeds:
        t.copyin_numpy(a)


def test_conv_bias_through_the_front_ends_reshape(B):
    """onnx.py:159-190: Conv, Reshape(bias, [1, F, 1, 1]), Add, Relu — operator order [Conv, Reshape, Add, Relu]. One launch;
    the Reshape of a weight is not launched at all."""
    h = B.GraphHandler(B.cpu_runtime())
    rng = np.random.default_rng(0)
    x = h.tensor([2, 32, 8, 8], F16)
    x.set_input()
    w, wa = weight(h, rng.standard_normal((64, 32, 3, 3)))
    b, ba = weight(h, rng.standard_normal((64,)))
    y = h.conv(x, w, None, 1, 1, 1, 1, 1, 1)
    r = h.reshape(b, None, [1, 64, 1, 1])
    out = h.relu(h.add(y, r, None), None)
    finish(h, [(x, rng.standard_normal((2, 32, 8, 8)).astype(np.float16)), (w, wa), (b, ba)])
    names = [str(o.op_type().id()).split(".")[-1] for o in h.operators()]
    assert names[:4] == ["Conv", "Reshape", "Add", "Relu"], names
    assert plan_of(h) == [(3, "conv+bias+relu", [0, 1, 2, 3])]


def test_matmul_add_bias_headsplit_in_the_exporters_order(B):
    """q = MatMul, Add — then k: MatMul, Add, Reshape, Transpose — v likewise — THEN q's Reshape, Transpose (HF BertSelfAttention):
    three chains of four operators, interleaved; planned as one grouped launch at the position of the last Transpose."""
    h = B.GraphHandler(B.cpu_runtime())
    rng = np.random.default_rng(1)
    Bt, S, NH, D = 2, 64, 2, 64
    H = NH * D
    x = h.tensor([Bt, S, H], F16)
    x.set_input()
    feeds = [(x, rng.standard_normal((Bt, S, H)).astype(np.float16))]
    lin = B.ActType.Linear

    def linear():
        w, wa = weight(h, rng.standard_normal((H, H)) / 8)
        b, ba = weight(h, rng.standard_normal((H,)))
        feeds.extend([(w, wa), (b, ba)])
        return h.add(b, h.matmul(x, w, None, False, False, None, lin, "default"), None)

    hd = lambda t: h.transpose(h.reshape(t, None, [Bt, S, NH, D]), None, [0, 2, 1, 3])
    ql = linear()
    k = hd(linear())
    v = hd(linear())
    q = hd(ql)
    outs = [h.abs(t, None) for t in (q, k, v)]
    finish(h, feeds)
    p = plan_of(h)
    fused = [it for it in p if it[1] != "op"]
    assert len(fused) == 1 and fused[0][1] == "matmul+bias+headsplit x3 (grouped)", p
    assert fused[0][0] == 11 and sorted(fused[0][2]) == list(range(12))


@pytest.mark.parametrize("kt", ["transpose", "merged", "transB"])
def test_attention_with_each_form_of_k_transposed(B, kt):
    h = B.GraphHandler(B.cpu_runtime())
    rng = np.random.default_rng(2)
    b, nh, s, d = 2, 2, 64, 64
    lin = B.ActType.Linear
    feeds = []

    def inp(shape):
        t = h.tensor(list(shape), F16)
        t.set_input()
        feeds.append((t, rng.standard_normal(shape).astype(np.float16)))
        return t

    q, v = inp((b, nh, s, d)), inp((b, nh, s, d))
    sc, sca = weight(h, np.array([8.0]))
    feeds.append((sc, sca))
    if kt == "merged":
        kx = h.transpose(inp((b, s, nh, d)), None, [0, 2, 3, 1])
        sm = h.matmul(q, kx, None, False, False, None, lin, "default")
    elif kt == "transpose":
        kx = h.transpose(inp((b, nh, s, d)), None, [0, 1, 3, 2])
        sm = h.matmul(q, kx, None, False, False, None, lin, "default")
    else:
        sm = h.matmul(q, inp((b, nh, s, d)), None, False, True, None, lin, "default")
    ctx = h.matmul(h.softmax(h.div(sm, sc, None), None, 3), v, None, False, False, None, lin, "default")
    out = h.reshape(h.transpose(ctx, None, [0, 2, 1, 3]), None, [b, s, nh * d])
    finish(h, feeds)
    whats = [w for _, w, _ in plan_of(h)]
    assert any(w.startswith("attention") for w in whats), whats  # (bridged through the workspace when O lands on K / V)
    if kt == "merged":  # K's view is transposed to head-major instead (same cost as the operator it replaces)
        assert "transpose(K head-major)" in whats
    assert "op" not in whats, whats  # every operator of the chain is inside a planned launch


def _bert(B, **kw):
    from model_bench import Builder, build_bert

    bl = Builder(B, B.cpu_runtime(), "f16", seed=0)
    build_bert(bl, 2, 128, 2, hidden=128, heads=2, ffn=256, vocab=100, **kw)
    bl.finish()
    return bl


def test_bert_in_front_end_form_plans_like_the_idealised_graph(B):
    """The graphs a user of the untouched front-end gets (MatMul + Add(bias), exporter order, Transpose(K)) must not plan
    worse than the friendlier lowering rounds 1-2 measured: same number of launches per layer, no lone bias Add."""
    counts = {}
    for name, kw in (("onnx", {}), ("merged_kt", {"merged_kt": True}), ("idealised", {"frontend": False}),
                     ("decomposed", {"decomposed": True})):
        bl = _bert(B, **kw)
        p = plan_of(bl.h)
        counts[name] = (len(bl.h.operators()), len(p), Counter(w for _, w, _ in p))
    for name in ("onnx", "merged_kt"):
        nops, nitems, c = counts[name]
        assert nitems <= counts["idealised"][1] + 1, (name, c, counts["idealised"][2])
        assert c["matmul+bias+headsplit x3 (grouped)"] == 2 and c["matmul+bias+gelu"] == 2, c
        assert sum(v for k, v in c.items() if k.startswith("attention")) == 2, c
        assert c["op"] <= 5, c  # gather, the position add + first norm, the classifier-less tail
    nops, nitems, c = counts["decomposed"]
    assert nops > counts["onnx"][0] + 40  # 9-operator LayerNorms, 5-operator Gelus
    assert nitems <= counts["onnx"][1] + 2, (c, counts["onnx"][2])
    assert sum(v for k, v in c.items() if "layernorm(decomposed)" in k or k == "layer_norm(decomposed)") == 5, c
    assert c["matmul+bias+gelu"] == 2, c


def test_resnet50_in_front_end_form(B):
    from model_bench import Builder, build_resnet50

    bl = Builder(B, B.cpu_runtime(), "f16", seed=0)
    build_resnet50(bl, 2, 64)
    bl.finish()
    names = [str(o.op_type().id()).split(".")[-1] for o in bl.h.operators()]
    assert names.count("Conv") == 53 and names.count("Reshape") == 53
    p = plan_of(bl.h)
    convs = [it for it in p if it[1].startswith("conv")]
    # (a chain is cut when the memory planner put its output on the conv's own input and the input cannot be bridged — a
    # strided layer; the conv then runs alone and Reshape + Add + Relu behind it)
    assert len(convs) >= 50 and all("+bias" in it[1] for it in convs), Counter(w for _, w, _ in p)
    launched_alone = Counter(names[m] for _, w, mem in p if w == "op" for m in mem)
    assert launched_alone["Reshape"] == launched_alone["Conv"] == 53 - len(convs), launched_alone


def test_llama_block_in_front_end_form(B):
    from model_bench import Builder, build_llama_block

    bl = Builder(B, B.cpu_runtime(), "f16", seed=0)
    build_llama_block(bl, 2, 128, heads=4, ffn=512)
    bl.finish()
    c = Counter(w for _, w, _ in plan_of(bl.h))
    assert c["attention"] == 1 and c["silu_mul(parked b)"] + c["silu_mul"] == 1 and c["rope+headsplit"] + c["rope+headsplit(parked x)"] == 2, c


def test_fusion_off_plans_one_item_per_operator(B, monkeypatch):
    import os
    import subprocess

    code = ("import sys; sys.path.insert(0, 'tests'); from conftest import load_backend_module; B = load_backend_module();"
            "h = B.GraphHandler(B.cpu_runtime()); x = h.tensor([4, 4], 1); h.relu(h.add(x, x, None), None); h.data_malloc();"
            "print(h.rocm_fusion_plan())")
    r = subprocess.run([sys.executable, "-c", code], cwd=Path(__file__).resolve().parent.parent, capture_output=True, text=True,
                       env=dict(os.environ, INFINI_ROCM_FUSION="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "0 op [0]" in r.stdout and "1 op [1]" in r.stdout


@pytest.mark.parametrize("kernel,channels,expect_res", [(1, 64, True), (3, 64, False), (1, 32, False)])
def test_residual_join_on_odd_planes_rides_only_in_a_pixel_slot_gemm(B, kernel, channels, expect_res):
    """conv -> reshape(bias) -> add -> add(residual) -> relu on a 7 x 7 plane (ResNet's last stage). The library takes a residual in
    the conv epilogue on odd planes only where the layer runs as a pixel-slot GEMM (1 x 1, C % 64 == 0, >= 128 filters:
    csrc/conv.hip, rocm_fusion.cc planConv); a 3 x 3 layer or a 32-channel one keeps conv+bias and a separate add+relu."""
    h = B.GraphHandler(B.cpu_runtime())
    rng = np.random.default_rng(11)
    x = h.tensor([2, channels, 7, 7], F16)
    x.set_input()
    res = h.tensor([2, 128, 7, 7], F16)
    res.set_input()
    w, wa = weight(h, rng.standard_normal((128, channels, kernel, kernel)))
    b, ba = weight(h, rng.standard_normal((128,)))
    y = h.conv(x, w, None, kernel // 2, kernel // 2, 1, 1, 1, 1)
    r = h.reshape(b, None, [1, 128, 1, 1])
    out = h.relu(h.add(h.add(y, r, None), res, None), None)
    finish(h, [(x, rng.standard_normal((2, channels, 7, 7)).astype(np.float16)), (res, rng.standard_normal((2, 128, 7, 7)).astype(np.float16)),
               (w, wa), (b, ba)])
    plan = plan_of(h)
    whats = [p[1] for p in plan]
    if expect_res:
        assert len(plan) == 1 and "res" in whats[0] and "relu" in whats[0], plan
    else:
        assert not any("res" in wt for wt in whats), plan
        conv_item = [pl for pl in plan if pl[1].startswith("conv+bias")]
        assert len(conv_item) == 1 and conv_item[0][2] == [0, 1, 2], plan  # the join (operators 3, 4) stays outside the conv launch


def test_forwarding_declines_when_another_branch_owns_the_convs_block(B):
    """Round-3 advisor finding: buffer forwarding writes the Conv operator's OWN buffer at the chain's slot, but the memory
    planner considers that buffer free once the bias Add has read it — another branch's operator between the Add and the
    chain's last member may have been given the block (tests/_fwd_graph.py builds exactly that; the placement is verified
    here through the CPU runtime's shared arena). The planner must not forward: the shorter chain conv + bias is planned."""
    import _fwd_graph as G

    h, t, feeds = G.build(B, B.cpu_runtime())

    def lands_on(a, b):  # does writing tensor a change tensor b (same arena block)?
        shape = t[a].shape()
        t[a].copyin_numpy(np.zeros(shape, np.float16))
        before = t[b].copyout_numpy().copy()
        t[a].copyin_numpy(np.full(shape, 7.0, np.float16))
        return float((before != t[b].copyout_numpy()).mean())

    assert lands_on("y", "v") > 0.9, "scenario: Tanh's output was planned onto the conv's own output block"
    assert lands_on("out", "x") > 0.4, "scenario: the chain's final output was planned onto the conv's input"
    assert lands_on("t", "x") == 0 and lands_on("v", "x") == 0
    plan = plan_of(h)
    assert not any("forwarded" in what for _, what, _ in plan), plan
    assert (7, "conv+bias", [6, 7]) in plan and (8, "op", [8]) in plan and (9, "op", [9]) in plan, plan
