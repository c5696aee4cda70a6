"""The drop-in test: the REFERENCE's graph executor (GraphHandler / GraphObj / KernelRegistry, compiled
from /root/reference) driving OUR Device::ROCM kernels through backend.RocmRuntime. Written like the
reference's own kernel tests: build the graph, copy inputs in, run, copy out, compare with the golden
vector — and, like its differential tests (test_cuda_unary.cc:12-41), with the same graph run on the
reference's native-CPU runtime in the same process."""
import numpy as np
import pytest
from conftest import kat

from oracle import ref_ops as R

pytestmark = pytest.mark.gpu
CU = "test/kernels/cuda/"
F32, F16, I64, I32, U8 = 1, 10, 7, 6, 2


@pytest.fixture(scope="module")
def B(plugin_backend):
    return plugin_backend


PLAN_ONLY = bool(__import__("os").environ.get("IROCM_TEST_PLAN_ONLY"))  # debugging aid: print the launch plans on a GPU-less box


@pytest.fixture(scope="module")
def rocm(B):
    return None if PLAN_ONLY else B.RocmRuntime(0)


def put(t, a):
    t.copyin_numpy(np.ascontiguousarray(a))


def get(t, shape=None):
    a = t.copyout_numpy()
    return a if shape is None else a.reshape(shape)


def build(B, runtime, fn, inputs):
    """inputs: list of (shape, dtype_code, array). fn(handler, tensors) -> output tensor(s)."""
    h = B.GraphHandler(runtime)
    ts = [h.tensor(list(s), d) for s, d, _ in inputs]
    out = fn(h, ts)
    h.data_malloc()
    for t, (_, _, a) in zip(ts, inputs):
        put(t, a)
    return h, out


def run_both(B, rocm, fn, inputs):
    h, out = build(B, rocm, fn, inputs)
    h.run()
    got = get(out)
    hc, outc = build(B, B.cpu_runtime(), fn, inputs)
    hc.run()
    return got, get(outc)


MATMUL = [("inc", "one", False, False, (1, 3, 5), (1, 5, 2), 50), ("inc", "inc", True, False, (2, 3, 4), (2, 3, 2), 53),
          ("inc", "inc", False, False, (2, 3, 5), (5, 2), 58), ("inc", "inc", True, False, (2, 5, 3), (5, 2), 61),
          ("inc", "inc", False, False, (3, 5), (5, 2), 65)]


@pytest.mark.parametrize("case", MATMUL)
def test_matmul_kats_through_reference_executor(B, rocm, case):
    ga, gb, ta, tb, sa, sb, line = case
    g = {"inc": R.incremental, "one": R.ones}
    h, c = build(B, rocm, lambda h, t: h.matmul(t[0], t[1], None, ta, tb, None, B.ActType.Linear, "default"),
                 [(sa, F32, g[ga](sa)), (sb, F32, g[gb](sb))])
    h.run()
    assert R.equal_data(get(c).ravel(), kat(CU + "test_cuda_matmul.cc", line, "float"), 1e-6)


def test_matmul_with_bias_and_fp16(B, rocm):
    rng = np.random.default_rng(0)
    a = rng.standard_normal((4, 128, 64)).astype(np.float16)
    w = rng.standard_normal((64, 256)).astype(np.float16)
    bias = rng.standard_normal((256,)).astype(np.float16)
    h, c = build(B, rocm, lambda h, t: h.matmul(t[0], t[1], None, False, False, t[2], B.ActType.Linear, "default"),
                 [(a.shape, F16, a), (w.shape, F16, w), (bias.shape, F16, bias)])
    h.run()
    want = R.matmul(a.astype(np.float64), w.astype(np.float64), bias.astype(np.float64))
    assert np.allclose(get(c).astype(np.float64).reshape(want.shape), want, rtol=2e-3, atol=3e-2)


@pytest.mark.parametrize("g,line", [(R.ones, 50), (R.incremental, 53)])
def test_conv_kats_through_reference_executor(B, rocm, g, line):
    h, y = build(B, rocm, lambda h, t: h.conv(t[0], t[1], None, 1, 1, 2, 1, 1, 2),
                 [((1, 3, 4, 4), F32, g((1, 3, 4, 4))), ((2, 3, 3, 3), F32, g((2, 3, 3, 3)))])
    h.run()
    assert R.equal_data(get(y).ravel(), kat(CU + "test_cuda_conv.cc", line, "float"), 1e-6)


@pytest.mark.parametrize("axis,line", [(0, 73), (1, 86), (2, 99), (3, 110)])
def test_softmax_kats_through_reference_executor(B, rocm, axis, line):
    x = np.arange(24, dtype=np.float32).reshape(2, 3, 2, 2)
    h, y = build(B, rocm, lambda h, t: h.softmax(t[0], None, axis), [(x.shape, F32, x)])
    h.run()
    assert R.equal_data(get(y).ravel(), kat(CU + "test_cuda_softmax.cc", line, "float"), 1e-6)


@pytest.mark.parametrize("si,yi,bi", [(157, 158, 165), (172, 173, 180), (187, 188, 195), (202, 203, None)])
def test_layernorm_kats_through_reference_executor(B, rocm, si, yi, bi):
    f = CU + "test_cuda_layernorm.cc"
    x = np.arange(36, dtype=np.float32).reshape(2, 3, 2, 3)
    scale = kat(f, si, "float").astype(np.float32)
    ins = [(x.shape, F32, x), (scale.shape, F32, scale)]
    if bi:
        bias = kat(f, bi, "float").astype(np.float32)
        ins.append((bias.shape, F32, bias))
    h, y = build(B, rocm, lambda h, t: h.layerNormalization(t[0], t[1], None, t[2] if bi else None, 1e-5, 3, 1), ins)
    h.run()
    assert R.equal_data(get(y).ravel(), kat(f, yi, "float"), 2e-6)


@pytest.mark.parametrize("name", ["relu", "silu", "abs", "sigmoid", "tanh", "hardSigmoid", "hardSwish", "sqrt", "neg", "erf", "gelu"])
@pytest.mark.parametrize("shape", [(1, 2, 2, 3), (13,), (2, 3, 4, 5, 6)])
def test_unary_device_vs_reference_native_cpu(B, rocm, name, shape):
    """test_cuda_unary.cc:122-143 verbatim in spirit: same op on the device and on the reference's CPU kernel."""
    x = R.incremental(shape)
    got, ref = run_both(B, rocm, lambda h, t: getattr(h, name)(t[0], None), [(shape, F32, x)])
    assert R.equal_data(got.ravel(), ref.ravel(), 2e-6), name


@pytest.mark.parametrize("name", ["add", "sub", "mul", "div"])
def test_binary_device_vs_reference_native_cpu(B, rocm, name):
    rng = np.random.default_rng(1)
    a = rng.uniform(0.5, 2, (2, 3, 4, 5)).astype(np.float32)
    b = rng.uniform(0.5, 2, (3, 1, 5)).astype(np.float32)
    got, ref = run_both(B, rocm, lambda h, t: getattr(h, name)(t[0], t[1], None), [(a.shape, F32, a), (b.shape, F32, b)])
    assert np.allclose(got, ref, rtol=1e-6, atol=1e-7)


def test_conv_pool_concat_transpose_vs_reference_native_cpu(B, rocm):
    rng = np.random.default_rng(2)
    x = np.abs(rng.standard_normal((2, 4, 9, 8))).astype(np.float32)
    w = rng.standard_normal((6, 4, 3, 3)).astype(np.float32)

    def net(h, t):
        y = h.conv(t[0], t[1], None, 1, 1, 1, 1, 1, 1)
        y = h.relu(y, None)
        p = h.maxPool(y, None, 3, 3, 1, 1, 1, 1, 2, 2, 0)
        q = h.avgPool(y, None, 3, 3, 1, 1, 1, 1, 2, 2, 0)
        c = h.concat([p, q], None, 1)
        return h.transpose(c, None, [0, 2, 3, 1])

    got, ref = run_both(B, rocm, net, [(x.shape, F32, x), (w.shape, F32, w)])
    assert got.shape == ref.shape or got.size == ref.size
    assert np.allclose(got.ravel(), ref.ravel(), rtol=1e-4, atol=1e-4)


def test_gather_int64_indices_bit_exact(B, rocm):
    table = np.random.default_rng(3).standard_normal((1000, 64)).astype(np.float32)
    ids = np.random.default_rng(4).integers(0, 1000, (4, 16)).astype(np.int64)
    h, y = build(B, rocm, lambda h, t: h.gather(t[0], t[1], None, 0), [(table.shape, F32, table), (ids.shape, I64, ids)])
    h.run()
    assert np.array_equal(get(y).reshape(4, 16, 64), table[ids])


def test_transformer_block_graph_eager_vs_hipgraph(B, rocm):
    """A BERT-style encoder block built op by op (MatMul, Add, Reshape, Transpose, Div, Softmax, LayerNorm,
    Gelu) through the reference GraphHandler; eager run == hipGraph replay == fp64 oracle."""
    rng = np.random.default_rng(5)
    Bt, S, Hd, NH = 2, 32, 64, 4
    D = Hd // NH
    x = rng.standard_normal((Bt, S, Hd)).astype(np.float32)
    W = {k: (rng.standard_normal((Hd, Hd)) / 8).astype(np.float32) for k in "qkvo"}
    W1 = (rng.standard_normal((Hd, 4 * Hd)) / 8).astype(np.float32)
    W2 = (rng.standard_normal((4 * Hd, Hd)) / 16).astype(np.float32)
    g = np.ones(Hd, np.float32)
    bt = np.zeros(Hd, np.float32)
    sc = np.array(np.sqrt(D), np.float32).reshape(1)
    ins = [(x.shape, F32, x)] + [(W[k].shape, F32, W[k]) for k in "qkvo"] + [(W1.shape, F32, W1), (W2.shape, F32, W2),
                                                                            (g.shape, F32, g), (bt.shape, F32, bt), ((1,), F32, sc)]
    lin = B.ActType.Linear

    def net(h, t):
        xin, wq, wk, wv, wo, w1, w2, gg, bb, scale = t
        mm = lambda a, b, tb=False: h.matmul(a, b, None, False, tb, None, lin, "default")
        heads = lambda y: h.transpose(h.reshape(y, None, [Bt, S, NH, D]), None, [0, 2, 1, 3])
        q, k, v = heads(mm(xin, wq)), heads(mm(xin, wk)), heads(mm(xin, wv))
        s = h.div(mm(q, k, True), scale, None)
        p = h.softmax(s, None, 3)
        ctx = h.reshape(h.transpose(mm(p, v), None, [0, 2, 1, 3]), None, [Bt, S, Hd])
        y = h.layerNormalization(h.add(xin, mm(ctx, wo), None), gg, None, bb, 1e-5, 2, 1)
        f = mm(h.gelu(mm(y, w1), None), w2)
        return h.layerNormalization(h.add(y, f, None), gg, None, bb, 1e-5, 2, 1)

    h = B.GraphHandler(rocm)
    ts = [h.tensor(list(s), d) for s, d, _ in ins]
    out = net(h, ts)
    h.data_malloc()

    def feed():  # the planner may reuse an input's memory for a later tensor: re-feed before every run
        for t, (_, _, a) in zip(ts, ins):
            put(t, a)

    feed()
    h.run()
    eager = get(out).copy()
    before = rocm.hip_graph_capture_count()
    feed()
    h.run_with_hipgraph()
    assert np.array_equal(get(out), eager)
    feed()
    h.run_with_hipgraph()
    assert rocm.hip_graph_capture_count() == before + 1  # captured once, replayed once
    assert np.array_equal(get(out), eager)
    # oracle
    X = x.astype(np.float64)
    hd = lambda y: y.reshape(Bt, S, NH, D).transpose(0, 2, 1, 3)
    q, k, v = hd(X @ W["q"]), hd(X @ W["k"]), hd(X @ W["v"])
    p = R.softmax(q @ k.transpose(0, 1, 3, 2) / np.sqrt(D), 3)
    ctx = (p @ v).transpose(0, 2, 1, 3).reshape(Bt, S, Hd)
    y = R.layer_norm(X + ctx @ W["o"], g, bt, 1e-5, 2)
    want = R.layer_norm(y + R.unary("gelu", y @ W1) @ W2, g, bt, 1e-5, 2)
    assert np.allclose(eager.reshape(want.shape), want, rtol=1e-4, atol=1e-4)


def test_hipgraph_cache_semantics(B):
    """test/cuda/test_cudagraph.cc: capture once, replay; a second graph gets its own capture; the LRU is bounded;
    new input CONTENTS in the same buffers need no recapture."""
    rt = B.RocmRuntime(0, 2)

    def make(n):
        h = B.GraphHandler(rt)
        a = h.tensor([n, n], F32)
        y = h.relu(h.add(a, a, None), None)
        h.data_malloc()
        return h, a, y

    h1, a1, y1 = make(8)
    put(a1, np.full((8, 8), -1, np.float32))
    h1.run_with_hipgraph()
    assert rt.hip_graph_capture_count() == 1 and np.all(get(y1) == 0)
    put(a1, np.full((8, 8), 2, np.float32))
    h1.run_with_hipgraph()
    assert rt.hip_graph_capture_count() == 1 and np.all(get(y1) == 4)  # replay sees the new contents
    h2, a2, y2 = make(16)
    put(a2, np.ones((16, 16), np.float32))
    h2.run_with_hipgraph()
    assert rt.hip_graph_capture_count() == 2 and rt.hip_graph_cache_size() == 2
    h3, a3, y3 = make(4)
    put(a3, np.ones((4, 4), np.float32))
    h3.run_with_hipgraph()
    assert rt.hip_graph_cache_size() == 2  # bounded LRU (capacity 2)
    put(a1, np.full((8, 8), 3, np.float32))  # (inputs are re-fed: the planner may alias y onto a)
    h1.run_with_hipgraph()  # evicted -> recaptured, still correct
    assert rt.hip_graph_capture_count() == 4 and np.all(get(y1) == 6)
    rt.clear_hip_graph_cache()
    assert rt.hip_graph_cache_size() == 0


def test_missing_kernel_is_a_loud_error(B, rocm):
    h = B.GraphHandler(rocm)
    x = h.tensor([2, 3, 4, 4], F32)
    h.resize(x, None, None, h.tensor([4], I64), None, None, [2, 3, 8, 8], "nearest", "floor", "half_pixel") if False else None
    # InstanceNormalization has no ROCM kernel registered: running it must raise, not silently fall back
    y = h.instanceNormalization(x, None, h.tensor([3], F32), h.tensor([3], F32), 1e-5)
    h.data_malloc()
    put(x, np.ones((2, 3, 4, 4), np.float32))
    with pytest.raises(RuntimeError):
        h.run()


def test_rope_through_reference_executor(B, rocm):
    """h.RoPE(pos, x) on Device::ROCM vs the oracle (head dim 128, theta 1e4 as rope.cc:25 / rope.cu:18), every
    (batch, position) rotated; first 32 columns of a ones/position-1 row reproduce test_cuda_rope.cc:29."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 9, 256)).astype(np.float32)
    x[0, 1, :] = 0
    x[0, 1, :32] = 1
    pos = np.tile(np.arange(9, dtype=np.uint32), (2, 1))
    U32 = 12
    h, out = build(B, rocm, lambda h, t: h.RoPE(t[0], t[1], None), [((2, 9), U32, pos), ((2, 9, 256), F32, x)])
    h.run()
    got = get(out, (2, 9, 256))
    assert np.allclose(got, R.rope(pos, x, 128), rtol=1e-4, atol=1e-5)
    assert R.equal_data(got[0, 1, :32], kat(CU + "test_cuda_rope.cc", 29, "float"), 2e-6)


def _resblock(h, t):
    """conv -> add(bias) -> relu -> conv -> add(bias) -> add(identity) -> relu, as onnx.py emits a folded-BN block."""
    y = h.relu(h.add(h.conv(t[0], t[1], None, 1, 1, 1, 1, 1, 1), t[2], None), None)
    z = h.add(h.conv(y, t[3], None, 1, 1, 1, 1, 1, 1), t[4], None)
    return h.relu(h.add(z, t[0], None), None)


@pytest.mark.parametrize("code,npdt", [(F32, np.float32), (F16, np.float16)])
def test_fusion_is_invisible(B, rocm, code, npdt):
    """Launch-time fusion (rocm_fusion.cc) vs one kernel per operator vs the reference's native-CPU runtime."""
    rng = np.random.default_rng(11)
    c = 32
    ins = [((2, c, 9, 10), code, rng.standard_normal((2, c, 9, 10)).astype(npdt)),
           ((c, c, 3, 3), code, (rng.standard_normal((c, c, 3, 3)) / 17).astype(npdt)),
           ((1, c, 1, 1), code, rng.standard_normal((1, c, 1, 1)).astype(npdt)),
           ((c, c, 3, 3), code, (rng.standard_normal((c, c, 3, 3)) / 17).astype(npdt)),
           ((1, c, 1, 1), code, rng.standard_normal((1, c, 1, 1)).astype(npdt))]
    got = {}
    try:
        for on in (True, False):
            rocm.set_fusion(on)
            assert rocm.get_fusion() == on
            h, out = build(B, rocm, _resblock, ins)
            before = rocm.fused_launch_count()
            h.run()
            assert (rocm.fused_launch_count() - before > 0) == on
            got[on] = get(out).astype(np.float64)
    finally:
        rocm.set_fusion(True)
    if code == F32:
        assert np.array_equal(got[True], got[False])  # same operations in the same order
        hc, outc = build(B, B.cpu_runtime(), _resblock, ins)
        hc.run()
        assert np.allclose(got[True], get(outc), rtol=1e-4, atol=1e-5)
    else:
        assert np.allclose(got[True], got[False], rtol=2e-3, atol=2e-3)
        x, w1, b1, w2, b2 = [a.astype(np.float64) for _, _, a in ins]
        y = np.maximum(R.conv2d(x, w1, 1, 1, 1, 1, 1, 1) + b1, 0)
        want = np.maximum(R.conv2d(y, w2, 1, 1, 1, 1, 1, 1) + b2 + x, 0)
        assert np.allclose(got[True], want, rtol=4e-3, atol=4e-3)


def test_matmul_gelu_fusion(B, rocm):
    """MatMul(+bias) -> Gelu (BERT's FFN up-projection): one GEMM launch with the Gelu in its epilogue (rocm_fusion.cc,
    gemm256p_kernel.h's compile-time instantiation at a persistent-kernel size, the run-time act elsewhere) vs two kernels
    vs the fp64 oracle. f32 is not fused (the matcher is f16 / bf16 only)."""
    rng = np.random.default_rng(33)
    for (m, k, n) in ((1024, 256, 1536), (96, 64, 80)):
        ins = [((m, k), F16, (rng.standard_normal((m, k)) * 1.5).astype(np.float16)),
               ((k, n), F16, (rng.standard_normal((k, n)) / 8).astype(np.float16)),
               ((n,), F16, rng.standard_normal((n,)).astype(np.float16))]
        def fn(h, t):
            for x in t:  # weights live outside the planner's recycled arena: the Gelu output cannot land on a dead operand
                x.set_weight()  # (in which case the matcher rightly refuses: the GEMM would overwrite what it still reads)
            return h.gelu(h.matmul(t[0], t[1], None, False, False, t[2], B.ActType.Linear, "default"), None)

        got = {}
        try:
            for on in (True, False):
                rocm.set_fusion(on)
                h, out = build(B, rocm, fn, ins)
                before = rocm.fused_launch_count()
                h.run()
                assert (rocm.fused_launch_count() - before == 1) == on
                got[on] = get(out).astype(np.float64)
        finally:
            rocm.set_fusion(True)
        a, w, bias = [x.astype(np.float64) for _, _, x in ins]
        want = R.unary("gelu", a @ w + bias)
        assert np.abs(a @ w + bias).max() > 6  # both tails of erf
        assert np.allclose(got[True], want, rtol=2e-3, atol=2e-3)
        assert np.allclose(got[True], got[False], rtol=2e-3, atol=2e-3)


def test_fusion_keeps_tensors_that_someone_else_reads(B, rocm):
    """The conv output feeds the bias-add AND a second consumer: it must be materialised (no fusion across it)."""
    rng = np.random.default_rng(12)
    ins = [((1, 32, 6, 6), F32, rng.standard_normal((1, 32, 6, 6)).astype(np.float32)),
           ((32, 32, 3, 3), F32, (rng.standard_normal((32, 32, 3, 3)) / 17).astype(np.float32)),
           ((1, 32, 1, 1), F32, rng.standard_normal((1, 32, 1, 1)).astype(np.float32))]

    def fn(h, t):
        y = h.conv(t[0], t[1], None, 1, 1, 1, 1, 1, 1)
        a = h.relu(h.add(y, t[2], None), None)
        return h.mul(a, y, None)

    got, want = run_both(B, rocm, fn, ins)
    assert np.allclose(got, want, rtol=1e-4, atol=1e-5)


def test_attention_chain_fusion(B, rocm):
    """MatMul(q, k^T) -> Div -> Add(mask) -> Softmax -> MatMul(p, v) as BERT emits it: one fused attention launch
    (rocm_fusion.cc) vs five kernels vs the fp64 oracle."""
    rng = np.random.default_rng(21)
    b, h, s, d = 2, 3, 96, 64
    mask = np.where(rng.random((b, 1, 1, s)) < 0.85, 0.0, -10000.0).astype(np.float16)
    ins = [((b, h, s, d), F16, rng.standard_normal((b, h, s, d)).astype(np.float16)),
           ((b, h, s, d), F16, rng.standard_normal((b, h, s, d)).astype(np.float16)),
           ((b, h, s, d), F16, rng.standard_normal((b, h, s, d)).astype(np.float16)),
           ((1,), F16, np.array([np.sqrt(d)], np.float16)),
           ((b, 1, 1, s), F16, mask)]

    def fn(hd, t):
        lin = B.ActType.Linear
        sc = hd.matmul(t[0], t[1], None, False, True, None, lin, "default")
        sc = hd.add(hd.div(sc, t[3], None), t[4], None)
        return hd.matmul(hd.softmax(sc, None, 3), t[2], None, False, False, None, lin, "default")

    got = {}
    try:
        for on in (True, False):
            rocm.set_fusion(on)
            hh, out = build(B, rocm, fn, ins)
            before = rocm.fused_launch_count()
            hh.run()
            assert rocm.fused_launch_count() - before == (1 if on else 0)  # the whole chain is ONE launch
            got[on] = get(out).astype(np.float64).reshape(b, h, s, d)
    finally:
        rocm.set_fusion(True)
    q, k, v = (a.astype(np.float64) for _, _, a in ins[:3])
    want = R.attention(q, k, v, 1.0 / float(np.float16(np.sqrt(d))), mask.astype(np.float64))
    assert np.allclose(got[True], want, rtol=3e-3, atol=3e-3)
    assert np.allclose(got[True], got[False], rtol=4e-3, atol=4e-3)


def test_first_hipgraph_run_sizes_the_workspace(B):
    """A graph whose kernels need scratch (conv_s1 re-packs weights into the workspace) captured on a FRESH runtime
    with no eager run before: the first capture fails on workspace growth, the runtime runs the graph once eagerly
    and captures again (rocm_runtime.cc::runWithHipGraph)."""
    rt = B.RocmRuntime(0)
    rng = np.random.default_rng(31)
    ins = [((2, 32, 12, 12), F16, rng.standard_normal((2, 32, 12, 12)).astype(np.float16)),
           ((48, 32, 3, 3), F16, (rng.standard_normal((48, 32, 3, 3)) / 17).astype(np.float16))]
    h, out = build(B, rt, lambda hd, t: hd.conv(t[0], t[1], None, 1, 1, 1, 1, 1, 1), ins)
    h.run_with_hipgraph()
    got = get(out).astype(np.float64).reshape(2, 48, 12, 12)
    assert rt.hip_graph_capture_count() == 1
    want = R.conv2d(ins[0][2].astype(np.float64), ins[1][2].astype(np.float64), 1, 1, 1, 1, 1, 1)
    assert np.allclose(got, want, rtol=3e-3, atol=3e-3)
    h.run_with_hipgraph()
    assert rt.hip_graph_capture_count() == 1 and np.allclose(get(out).astype(np.float64).reshape(2, 48, 12, 12), want, rtol=3e-3, atol=3e-3)


def test_attention_kvcache_through_reference_executor(B, rocm):
    """test_cuda_attention.cc:10-43 on Device::ROCM, plus a second decode step at position 3 vs the oracle."""
    U32 = 12
    one = np.ones((1, 1, 1, 128), np.float32)
    ins = [((1, 1, 1, 128), F32, np.zeros((1, 1, 1, 128), np.float32)), ((1, 1, 1, 128), F32, np.zeros((1, 1, 1, 128), np.float32)),
           ((1, 1, 1, 128), F32, one), ((1, 1, 1, 128), F32, one), ((1, 1, 1, 128), F32, one), ((1, 1), U32, np.zeros((1, 1), np.uint32))]
    h, out = build(B, rocm, lambda hd, t: hd.attentionKVCache(t[0], t[1], t[2], t[3], t[4], t[5], None), ins)
    h.run()
    assert R.equal_data(get(out).ravel(), kat(CU + "test_cuda_attention.cc", 36, "float"), 1e-6)
    rng = np.random.default_rng(9)
    b, hh, ms, d, pos = 2, 2, 8, 128, 3
    arrs = [rng.standard_normal(s).astype(np.float32) for s in [(b, hh, ms, d), (b, hh, ms, d), (b, hh, 1, d), (b, hh, 1, d), (b, hh, 1, d)]]
    ins = [(a.shape, F32, a) for a in arrs] + [((1, 1), U32, np.full((1, 1), pos, np.uint32))]
    h, out = build(B, rocm, lambda hd, t: hd.attentionKVCache(t[0], t[1], t[2], t[3], t[4], t[5], None), ins)
    h.run()
    want, _, _ = R.attention_kvcache(*arrs, pos)
    assert np.allclose(get(out).reshape(b, hh, 1, d), want, rtol=1e-5, atol=1e-5)


def test_gather_elements_extend_depth_to_space_through_reference_executor(B, rocm):
    """The three glue kernels added for SURVEY 8f-4, driven by the reference's operator objects."""
    GE = CU + "test_cuda_gather_elements.cc"
    h, out = build(B, rocm, lambda hd, t: hd.gatherElements(t[0], t[1], None, 0),
                   [((3, 3), I32, kat(GE, 19).astype(np.int32).reshape(3, 3)), ((2, 3), I64, kat(GE, 20).astype(np.int64).reshape(2, 3))])
    h.run()
    assert get(out).ravel().tolist() == [4, 8, 3, 7, 2, 3]  # test_cuda_gather_elements.cc:24
    # Extend has no GraphHandler binding in the reference (ffi_infinitensor.cc): covered through the C ABI in
    # tests/test_gpu_movement.py::test_extend_kat_and_depth_to_space
    a = np.random.default_rng(4).standard_normal((1, 8, 3, 5)).astype(np.float32)
    h, out = build(B, rocm, lambda hd, t: hd.depthToSpace(t[0], None, 2, "DCR"), [((1, 8, 3, 5), F32, a)])
    h.run()
    assert np.array_equal(get(out).reshape(1, 2, 6, 10), R.depth_to_space(a, 2, "DCR"))


def test_resize_through_reference_executor(B, rocm):
    """Three of the reference's Resize tests driven through its own operator object (ResizeObj derives scales / roi from
    the sizes / scales / roi tensors on Device::ROCM): nearest sizes, linear scales align_corners, cubic sizes."""
    RS = CU + "test_cuda_resize.cc"

    def run(x, sizes, scales, roi, mode, coord, nearest="round_prefer_floor"):
        h = B.GraphHandler(rocm)
        tx = h.tensor(list(x.shape), F32)
        ts = h.tensor([len(sizes)], I64) if sizes else None
        tc = h.tensor([len(scales)], F32) if scales else None
        tr = h.tensor([len(roi)], F32) if roi else None
        out = h.resize(tx, None, None, ts, tc, tr, sizes or [], scales or [], roi or [], mode, "stretch", nearest, coord)
        h.data_malloc()
        put(tx, x)
        h.run()
        return get(out).ravel()

    x = kat(RS, 16, "float").astype(np.float32).reshape(1, 1, 2, 4)
    assert R.equal_data(run(x, [1, 1, 1, 3], None, None, "nearest", "half_pixel"), kat(RS, 35, "float"), 1e-6)
    x = kat(RS, 336, "float").astype(np.float32).reshape(1, 1, 2, 4)
    assert R.equal_data(run(x, None, [1, 1, 0.6, 0.6], None, "linear", "align_corners"), kat(RS, 355, "float"), 1e-6)
    x = kat(RS, 759, "float").astype(np.float32).reshape(1, 1, 4, 4)
    assert R.equal_data(run(x, [1, 1, 9, 10], None, None, "cubic", "half_pixel"), kat(RS, 778, "float"), 1e-5)


def test_conv_transpose_through_reference_executor(B, rocm):
    """test_cuda_conv_transposed_2d.cc:101-135 on Device::ROCM."""
    h, out = build(B, rocm, lambda hd, t: hd.convTransposed2d(t[0], t[1], None, 0, 0, 1, 1, 1, 1, 0, 0),
                   [((1, 2, 3, 3), F32, R.incremental((1, 2, 3, 3))), ((2, 2, 3, 3), F32, R.incremental((2, 2, 3, 3)))])
    h.run()
    assert R.equal_data(get(out).ravel(), kat(CU + "test_cuda_conv_transposed_2d.cc", 129, "float"), 1e-6)


def test_add_layernorm_fusion(B, rocm):
    """Add -> LayerNormalization (BERT's residual join) as one launch; equal to the two-kernel chain up to rounding ties."""
    rng = np.random.default_rng(61)
    ins = [((4, 10, 96), F16, rng.standard_normal((4, 10, 96)).astype(np.float16)),
           ((4, 10, 96), F16, rng.standard_normal((4, 10, 96)).astype(np.float16)),
           ((96,), F16, rng.standard_normal((96,)).astype(np.float16)), ((96,), F16, rng.standard_normal((96,)).astype(np.float16))]

    def fn(hd, t):
        y = hd.layerNormalization(hd.add(t[0], t[1], None), t[2], None, t[3], 1e-5, 2, 1)
        return hd.mul(y, t[0], None)  # keeps t[0] alive past the pair

    got = {}
    try:
        for on in (True, False):
            rocm.set_fusion(on)
            hh, out = build(B, rocm, fn, ins)
            before = rocm.fused_launch_count()
            hh.run()
            assert rocm.fused_launch_count() - before == (1 if on else 0)
            got[on] = get(out).astype(np.float64)
    finally:
        rocm.set_fusion(True)
    assert np.allclose(got[True], got[False], rtol=2.0 ** -9, atol=1e-5)


def test_relu_maxpool_fusion_is_bit_exact(B, rocm):
    """Relu -> MaxPool (ResNet stem) as one launch; max and relu commute, so fused == unfused bit for bit
    (both the specialised 3x3/2 kernel and the generic one). The input stays live (second consumer), otherwise the
    planner puts the pool output on top of it and the rule must not fire."""
    rng = np.random.default_rng(71)
    for shape, k, pad, st in (((2, 8, 16, 16), 3, 1, 2), ((1, 3, 9, 7), 2, 0, 1)):
        x = rng.standard_normal(shape).astype(np.float16)
        got = {}

        def fn(hd, t):
            y = hd.maxPool(hd.relu(t[0], None), None, k, k, 1, 1, pad, pad, st, st, 0)
            return [y, hd.neg(t[0], None)]

        try:
            for on in (True, False):
                rocm.set_fusion(on)
                hh, outs = build(B, rocm, fn, [(shape, F16, x)])
                before = rocm.fused_launch_count()
                hh.run()
                assert rocm.fused_launch_count() - before == (1 if on else 0)
                got[on] = get(outs[0])
        finally:
            rocm.set_fusion(True)
        assert np.array_equal(got[True], got[False])
        want = R.pool2d(np.maximum(x.astype(np.float64), 0), "max", k, k, 1, 1, pad, pad, st, st, 0)
        assert np.array_equal(got[True].reshape(want.shape).astype(np.float64), want)


def test_tune_selects_kernel_variants_and_keeps_results(B, rocm, tmp_path):
    """h.tune() (reference: RuntimeObj::run(graph, tune = true)) on Device::ROCM: MatMul and Conv time their kernel
    variants (RocmTunableKernel, the role of the cuBLAS / cuDNN algo sweep in matmul.cc:187-208 / conv.cc:176-244) and
    leave type-3 / type-4 records in the PerfEngine; runs after that — eager, fused and hipGraph — launch the recorded
    variant and agree with the untuned result (every variant accumulates in fp32) and with the oracle; the records
    survive the JSON round trip."""
    import json

    rng = np.random.default_rng(21)
    x = rng.standard_normal((4, 64, 28, 28)).astype(np.float16)
    w = (rng.standard_normal((128, 64, 1, 1)) / 8).astype(np.float16)
    w3 = (rng.standard_normal((128, 128, 3, 3)) / 34).astype(np.float16)
    bs = rng.standard_normal((1, 128, 1, 1)).astype(np.float16)
    a = rng.standard_normal((512, 768)).astype(np.float16)
    m = (rng.standard_normal((768, 768)) / 28).astype(np.float16)
    lin = B.ActType.Linear

    def net(h, t):
        y = h.relu(h.add(h.conv(t[0], t[1], None, 0, 0, 1, 1, 1, 1), t[3], None), None)
        y = h.relu(h.add(h.conv(y, t[2], None, 1, 1, 1, 1, 1, 1), t[3], None), None)
        return y, h.matmul(t[4], t[5], None, False, False, None, lin, "default")

    ins = [(x.shape, F16, x), (w.shape, F16, w), (w3.shape, F16, w3), (bs.shape, F16, bs), (a.shape, F16, a), (m.shape, F16, m)]
    B.RocmRuntime.clear_perf()
    try:
        h, (yc, ym) = build(B, rocm, net, ins)
        h.run()
        c0, m0 = get(yc).astype(np.float64), get(ym).astype(np.float64)
        h.tune()
        assert B.RocmRuntime.perf_size() >= 5  # 2 conv workloads, 1 matmul, add, relu
        p = tmp_path / "perf.json"
        B.RocmRuntime.save_perf(str(p))
        recs = json.loads(p.read_text())["data"]
        types = sorted(r["type"] for _, r in recs)
        assert types.count(3) == 1 and types.count(4) == 2, types
        for _, r in recs:
            if r["type"] in (3, 4):
                assert -1 <= r["data"][0] <= 6 and r["data"][1] > 0
        results = []
        for mode in ("eager", "hipgraph"):
            hh, (yc2, ym2) = build(B, rocm, net, ins)
            (hh.run_with_hipgraph if mode == "hipgraph" else hh.run)()
            results.append((get(yc2).astype(np.float64), get(ym2).astype(np.float64)))
        # force every non-default variant through the record path as well: same sums whatever tune() picked
        for conv_v, mm_v in ((1, 0), (2, 1), (3, 3), (2, 2), (1, 4), (2, 5), (3, 6)):
            forced = {"data": [[k, {"type": r["type"], "data": [conv_v if r["type"] == 4 else mm_v, r["data"][1]]}
                                if r["type"] in (3, 4) else r] for k, r in recs]}
            p.write_text(json.dumps(forced))
            B.RocmRuntime.load_perf(str(p))
            hh, (yc2, ym2) = build(B, rocm, net, ins)
            hh.run()
            results.append((get(yc2).astype(np.float64), get(ym2).astype(np.float64)))
        X, W, W3, BS = (t.astype(np.float64) for t in (x, w, w3, bs))
        y1 = np.maximum(R.conv2d(X, W, 0, 0, 1, 1, 1, 1) + BS, 0)
        y1 = R.round_to(y1, "f16").astype(np.float64)
        want_c = np.maximum(R.conv2d(y1, W3, 1, 1, 1, 1, 1, 1) + BS, 0)
        want_m = a.astype(np.float64) @ m.astype(np.float64)
        for c1, m1 in results:
            assert np.allclose(c1.reshape(c0.shape), c0, rtol=2e-3, atol=2e-3)
            assert np.allclose(m1.reshape(m0.shape), m0, rtol=2e-3, atol=2e-3)
            assert np.allclose(c1.reshape(want_c.shape), want_c, rtol=4e-3, atol=4e-3)
            assert np.allclose(m1.reshape(want_m.shape), want_m, rtol=4e-3, atol=4e-3)
    finally:
        B.RocmRuntime.clear_perf()


def test_producer_writes_into_reshape_output(B, rocm):
    """producer -> Reshape-family copy (rocm_fusion.cc::tryLaunchIntoReshape): MatMul(+bias) -> Reshape -> Transpose ->
    Reshape and Gelu -> Flatten launch the producer with its output redirected into the copy's buffer — bit-identical to
    one kernel per operator (the same kernels run, only the destination differs) and equal to NumPy; a producer whose
    output has a second reader keeps its own buffer (no fusion)."""
    rng = np.random.default_rng(31)
    Bt, S, NH, D = 2, 48, 4, 32
    x = rng.standard_normal((Bt * S, NH * D)).astype(np.float32)
    w = (rng.standard_normal((NH * D, NH * D)) / 11).astype(np.float32)
    b = rng.standard_normal((NH * D,)).astype(np.float32)
    lin = B.ActType.Linear

    def chain(h, t):
        y = h.matmul(t[0], t[1], None, False, False, t[2], lin, "default")      # -> reshape: fused
        y = h.transpose(h.reshape(y, None, [Bt, S, NH, D]), None, [0, 2, 1, 3])  # transpose -> reshape: fused
        y = h.reshape(y, None, [Bt * NH, S * D])
        return h.flatten(h.gelu(y, None), None, 1)                               # gelu -> flatten: fused

    def shared(h, t):
        y = h.matmul(t[0], t[1], None, False, False, t[2], lin, "default")
        r = h.reshape(y, None, [Bt, S, NH * D])                                  # y is also read by the add: not fused
        return h.add(h.reshape(r, None, [Bt * S, NH * D]), y, None)

    ins = [(x.shape, F32, x), (w.shape, F32, w), (b.shape, F32, b)]
    want_chain = (x.astype(np.float64) @ w + b).reshape(Bt, S, NH, D).transpose(0, 2, 1, 3).reshape(Bt * NH, S * D)
    want_chain = R.unary("gelu", want_chain)
    want_shared = 2 * (x.astype(np.float64) @ w + b)

    def build_persistent(fn):  # operands that outlive the graph (weights): their storage is never recycled
        h = B.GraphHandler(rocm)
        ts = [h.tensor(list(s_), d_) for s_, d_, _ in ins]
        for t in ts:
            t.set_weight()
        out = fn(h, ts)
        h.data_malloc()
        for t, (_, _, a) in zip(ts, ins):
            put(t, a)
        return h, out

    got = {}
    try:
        for on in (True, False):
            rocm.set_fusion(on)
            for name, fn in (("chain", chain), ("shared", shared)):
                h, out = build_persistent(fn)
                before = rocm.fused_launch_count()
                h.run()
                fused = rocm.fused_launch_count() - before
                if name == "chain":
                    # MatMul -> Reshape always qualifies here (its operands are persistent); the later pairs only when
                    # the planner did not hand the copy's output the storage of the producer's dying input
                    assert (1 <= fused <= 3) if on else fused == 0, fused
                else:
                    assert fused == 0, fused  # the matmul has two readers; Reshape is not a producer of the rule
                got[(name, on)] = get(out).astype(np.float64)
    finally:
        rocm.set_fusion(True)
    assert np.array_equal(got[("chain", True)], got[("chain", False)])
    assert np.array_equal(got[("shared", True)], got[("shared", False)])
    assert np.allclose(got[("chain", True)].reshape(want_chain.shape), want_chain, rtol=1e-4, atol=1e-4)
    assert np.allclose(got[("shared", True)].reshape(want_shared.shape), want_shared, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("code,npdt", [(F16, np.float16), (F32, np.float32)])
def test_head_split_fusion_is_bit_identical(B, rocm, code, npdt):
    """MatMul(+bias) -> Reshape([B, S, H, D]) -> Transpose(0, 2, 1, 3) as one GEMM with a head-split store
    (rocm_fusion.cc::tryLaunchHeadSplit): identical bits to the three-kernel chain, for a rank-2 and a rank-3 MatMul."""
    rng = np.random.default_rng(41)
    Bt, S, NH, D = 2, 64, 3, 32
    w = (rng.standard_normal((NH * D, NH * D)) / 9).astype(npdt)
    b = rng.standard_normal((NH * D,)).astype(npdt)
    lin = B.ActType.Linear
    for xshape in ((Bt * S, NH * D), (Bt, S, NH * D)):
        x = rng.standard_normal(xshape).astype(npdt)

        def fn(h, t):
            y = h.matmul(t[0], t[1], None, False, False, t[2], lin, "default")
            return h.transpose(h.reshape(y, None, [Bt, S, NH, D]), None, [0, 2, 1, 3])

        got = {}
        try:
            for on in (True, False):
                rocm.set_fusion(on)
                h = B.GraphHandler(rocm)
                ts = [h.tensor(list(a.shape), code) for a in (x, w, b)]
                for t in ts:
                    t.set_weight()
                out = fn(h, ts)
                h.data_malloc()
                for t, a in zip(ts, (x, w, b)):
                    put(t, a)
                before = rocm.fused_launch_count()
                h.run()
                assert rocm.fused_launch_count() - before == (1 if on else 0)
                got[on] = get(out)
        finally:
            rocm.set_fusion(True)
        assert np.array_equal(got[True], got[False])
        want = (x.astype(np.float64).reshape(Bt * S, NH * D) @ w.astype(np.float64) + b).reshape(Bt, S, NH, D).transpose(0, 2, 1, 3)
        tol = 1e-4 if npdt is np.float32 else 4e-3
        assert np.allclose(got[True].astype(np.float64).reshape(want.shape), want, rtol=tol, atol=tol)


def test_grouped_qkv_head_split_is_bit_identical(B, rocm):
    """Three head-split projections of ONE activation (q, k, v) whose weights / biases / outputs sit at a uniform spacing
    run as one grouped GEMM launch (rocm_fusion.cc::tryLaunchHeadSplit: group index = batch index, zero A stride) —
    identical bits to nine kernels; a fourth consumer of the activation (a different width) stays outside the group."""
    rng = np.random.default_rng(43)
    Bt, S, NH, D = 2, 128, 4, 64
    hid = NH * D
    x = rng.standard_normal((Bt, S, hid)).astype(np.float16)
    ws = [(rng.standard_normal((hid, hid)) / 16).astype(np.float16) for _ in range(3)]
    bs = [rng.standard_normal((hid,)).astype(np.float16) for _ in range(3)]
    arrays = [x]
    for w, b in zip(ws, bs):  # a layer's parameters in creation order: w_q, b_q, w_k, b_k, w_v, b_v
        arrays += [w, b]
    lin = B.ActType.Linear
    got, launches = {}, {}
    try:
        for on in (True, False):
            rocm.set_fusion(on)
            h = B.GraphHandler(rocm)
            ts = [h.tensor(list(a.shape), F16) for a in arrays]
            for t in ts:
                t.set_weight()
            outs = []
            for j in range(3):
                y = h.matmul(ts[0], ts[1 + 2 * j], None, False, False, ts[2 + 2 * j], lin, "default")
                outs.append(h.transpose(h.reshape(y, None, [Bt, S, NH, D]), None, [0, 2, 1, 3]))
            h.data_malloc()
            for t, a in zip(ts, arrays):
                put(t, a)
            before = rocm.fused_launch_count()
            h.run()
            launches[on] = rocm.fused_launch_count() - before
            got[on] = [get(o) for o in outs]
    finally:
        rocm.set_fusion(True)
    assert launches[False] == 0 and launches[True] in (1, 2, 3)  # 1 = all three grouped (the planner's layout permitting)
    for j in range(3):
        assert np.array_equal(got[True][j], got[False][j])
        want = (x.astype(np.float64).reshape(Bt * S, hid) @ ws[j].astype(np.float64) + bs[j]).reshape(Bt, S, NH, D).transpose(0, 2, 1, 3)
        assert np.allclose(got[True][j].astype(np.float64), want, rtol=4e-3, atol=4e-3)


def test_grouped_plain_matmuls_are_hoisted_safely(B, rocm):
    """A decoder block's gate / up pattern — mm_g, Silu, mm_u, Mul — and three projections ahead of element-wise ops: the
    MatMuls of one activation run as ONE grouped launch, the later members ahead of their place in the list
    (rocm_fusion.cc::tryLaunchGroupedMatmul). Identical bits to the per-operator run; the operators that were jumped over
    still see their own inputs."""
    rng = np.random.default_rng(47)
    T, H, Fd = 512, 256, 768
    x = rng.standard_normal((T, H)).astype(np.float16)
    wg, wu = [(rng.standard_normal((H, Fd)) / 16).astype(np.float16) for _ in range(2)]
    wq, wk, wv = [(rng.standard_normal((H, H)) / 16).astype(np.float16) for _ in range(3)]
    arrays = [x, wg, wu, wq, wk, wv]
    lin = B.ActType.Linear
    got, launches = {}, {}
    try:
        for on in (True, False):
            rocm.set_fusion(on)
            h = B.GraphHandler(rocm)
            ts = [h.tensor(list(a.shape), F16) for a in arrays]
            for t in ts[1:]:
                t.set_weight()
            mm = lambda a, w: h.matmul(a, w, None, False, False, None, lin, "default")
            xin = h.relu(ts[0], None)  # the shared activation is itself an intermediate
            # The raw products stay alive to the end of the graph (they are outputs too): otherwise the planner hands mm_g's
            # buffer to mm_u as soon as Silu has read it, and the rule rightly refuses to run mm_u ahead of Silu.
            g_raw = mm(xin, ts[1])
            gated = h.mul(h.silu(g_raw, None), mm(xin, ts[2]), None)
            q_raw, k_raw = mm(xin, ts[3]), mm(xin, ts[4])
            q, k, v = h.neg(q_raw, None), h.abs(k_raw, None), h.sigmoid(mm(xin, ts[5]), None)
            outs = [gated, q, k, v, g_raw, q_raw, k_raw]
            h.data_malloc()
            for t, a in zip(ts, arrays):
                put(t, a)
            before = rocm.fused_launch_count()
            h.run()
            launches[on] = rocm.fused_launch_count() - before
            got[on] = [get(o) for o in outs]
    finally:
        rocm.set_fusion(True)
    assert launches[False] == 0 and launches[True] >= 1  # at least the gate / up pair (two members always sit "uniformly")
    for a, b in zip(got[True], got[False]):
        assert np.array_equal(a, b)
    xr = np.maximum(x.astype(np.float64), 0)
    silu = lambda t: t / (1 + np.exp(-t))
    r16 = lambda t: t.astype(np.float16).astype(np.float64)
    want = r16(silu(r16(xr @ wg.astype(np.float64)))) * r16(xr @ wu.astype(np.float64))
    assert np.allclose(got[True][0].astype(np.float64), want, rtol=6e-3, atol=6e-3)


def test_grouped_matmul_member_parked_in_the_workspace(B, rocm):
    """The same gate / up and q / k patterns as a real graph has them: the raw products are NOT kept alive, so the planner puts
    mm_u's output where mm_g's was (dead once Silu has read it). The group still runs as one launch: the second member's
    result is parked in the workspace and its one consumer (Mul / Abs, the next operator) reads it from there
    (`parked_member_count`). Identical bits to the per-operator run."""
    rng = np.random.default_rng(53)
    T, H, Fd = 4096, 256, 2816  # big enough that the grouped GEMM cannot be a split-K one (which would want the workspace itself)
    x = rng.standard_normal((T, H)).astype(np.float16)
    wg, wu = [(rng.standard_normal((H, Fd)) / 16).astype(np.float16) for _ in range(2)]
    wq, wk = [(rng.standard_normal((H, Fd)) / 16).astype(np.float16) for _ in range(2)]
    arrays = [x, wg, wu, wq, wk]
    lin = B.ActType.Linear
    got, parked = {}, {}
    try:
        for on in (True, False):
            rocm.set_fusion(on)
            h = B.GraphHandler(rocm)
            ts = [h.tensor(list(a.shape), F16) for a in arrays]
            for t in ts[1:]:
                t.set_weight()
            mm = lambda a, w: h.matmul(a, w, None, False, False, None, lin, "default")
            xin = h.relu(ts[0], None)
            gated = h.mul(h.silu(mm(xin, ts[1]), None), mm(xin, ts[2]), None)
            q, k = h.neg(mm(xin, ts[3]), None), h.abs(mm(xin, ts[4]), None)
            qk = h.add(q, k, None)
            outs = [gated, qk]
            h.data_malloc()
            for t, a in zip(ts, arrays):
                put(t, a)
            before = rocm.parked_member_count()
            h.run()
            first = [get(o) for o in outs]
            put(ts[0], x)  # the planner recycles the input's buffer: feed it again before the second run
            h.run_with_hipgraph()  # the parked pointer is part of the captured graph
            parked[on] = rocm.parked_member_count() - before
            got[on] = [get(o) for o in outs]
            for a, b in zip(first, got[on]):
                assert np.array_equal(a, b)
    finally:
        rocm.set_fusion(True)
    assert parked[False] == 0
    for a, b in zip(got[True], got[False]):
        assert np.array_equal(a, b)
    xr = np.maximum(x.astype(np.float64), 0)
    r16 = lambda t: t.astype(np.float16).astype(np.float64)
    want = r16(-r16(xr @ wq.astype(np.float64))) + r16(np.abs(r16(xr @ wk.astype(np.float64))))
    assert np.allclose(got[True][1].astype(np.float64), want, rtol=6e-3, atol=6e-3)
    assert parked[True] >= 1  # the planner recycles mm_g's buffer for mm_u here: the parked path is what ran


# ---- hipGraph cache: the remaining cases of test/cuda/test_cudagraph.cc, through backend.RocmRuntime -----------------
class _GraphFixture:
    """CudaGraphFixture of test_cudagraph.cc:29-72: input [batch, 2] @ identity weight [2, 2] -> Relu."""

    def __init__(self, B, rt, batch=8):
        self.B, self.rt = B, rt
        self.h = B.GraphHandler(rt)
        self.input = self.h.tensor([batch, 2], F32)
        self.weight = self.h.tensor([2, 2], F32)
        self.input.set_input()
        self.weight.set_weight()
        mm = self.h.matmul(self.input, self.weight, None, False, False, None, B.ActType.Linear, "default")
        self.output = self.h.relu(mm, None)
        self.output.set_output()
        self.h.data_malloc()
        put(self.weight, np.eye(2, dtype=np.float32))

    def reshape(self, batch):
        if self.input.shape() == [batch, 2]:
            return
        self.h.change_shape([batch, 2], self.input.fuid())
        self.h.shape_infer()
        self.h.data_malloc()

    def run(self, values):
        put(self.input, np.asarray(values, np.float32).reshape(self.input.shape()))
        self.h.run_with_hipgraph()

    def out(self):
        return get(self.output).ravel()


def _inc(batch, offset=0.0):
    return np.arange(batch * 2, dtype=np.float32) + offset


def test_hipgraph_first_run_captures_then_replays(B):
    """test_cudagraph.cc:80-98."""
    rt = B.RocmRuntime(0)
    fx = _GraphFixture(B, rt)
    fx.run(_inc(8, -4))
    assert rt.hip_graph_capture_count() == 1 and rt.hip_graph_cache_size() == 1
    assert np.array_equal(fx.out(), np.maximum(_inc(8, -4), 0))
    fx.run(_inc(8, 1))
    assert rt.hip_graph_capture_count() == 1 and np.array_equal(fx.out(), _inc(8, 1))


def test_hipgraph_reuses_previous_shape_in_same_storage(B):
    """test_cudagraph.cc:100-115: 8 -> 4 -> 8 rows in the same arena: the third run replays the first capture."""
    rt = B.RocmRuntime(0)
    fx = _GraphFixture(B, rt, 8)
    fx.run(_inc(8))
    fx.reshape(4)
    fx.run(_inc(4))
    assert rt.hip_graph_capture_count() == 2
    fx.reshape(8)
    fx.run(_inc(8, 10))
    assert rt.hip_graph_capture_count() == 2 and rt.hip_graph_cache_size() == 2
    assert np.array_equal(fx.out(), _inc(8, 10))


def test_hipgraph_ignores_tensor_contents(B):
    """test_cudagraph.cc:117-127: new weight VALUES in the same buffer need no recapture."""
    rt = B.RocmRuntime(0)
    fx = _GraphFixture(B, rt, 2)
    fx.run([1, -2, 3, -4])
    put(fx.weight, np.array([[2, 0], [0, 3]], np.float32))
    fx.run([-1, 2, 3, -4])
    assert rt.hip_graph_capture_count() == 1 and np.array_equal(fx.out(), [0, 6, 6, 0])


def test_hipgraph_invalidates_replaced_storage(B):
    """test_cudagraph.cc:146-168: growing past the arena replaces the storage -> every capture of the graph is dropped;
    trim_memory drops them too."""
    rt = B.RocmRuntime(0)
    fx = _GraphFixture(B, rt, 8)
    fx.run(_inc(8))
    assert rt.hip_graph_cache_size() == 1
    fx.reshape(1024)
    assert rt.hip_graph_cache_size() == 0
    fx.run(_inc(1024))
    assert rt.hip_graph_capture_count() == 2
    fx.reshape(8)
    fx.run(_inc(8))
    assert rt.hip_graph_capture_count() == 3
    fx.h.trim_memory()
    assert rt.hip_graph_cache_size() == 0
    fx.run(_inc(8, 3))
    assert rt.hip_graph_capture_count() == 4 and np.array_equal(fx.out(), _inc(8, 3))


def test_hipgraph_invalidates_topology_changes(B):
    """test_cudagraph.cc:170-186: adding a tensor to the graph invalidates its captures."""
    rt = B.RocmRuntime(0)
    fx = _GraphFixture(B, rt, 2)
    fx.run(_inc(2))
    assert rt.hip_graph_cache_size() == 1
    extra = fx.h.tensor([1], F32)
    extra.set_input()
    assert rt.hip_graph_cache_size() == 0
    fx.h.data_malloc()
    put(extra, np.ones(1, np.float32))
    fx.run(_inc(2, 2))
    assert rt.hip_graph_capture_count() == 2 and rt.hip_graph_cache_size() == 1
    assert np.array_equal(fx.out(), _inc(2, 2))


def test_hipgraph_bounded_lru_by_shape(B):
    """test_cudagraph.cc:188-211: capacity 2, shapes 8 / 6 / 4 of one graph."""
    rt = B.RocmRuntime(0, 2)
    fx = _GraphFixture(B, rt, 8)
    fx.run(_inc(8))
    fx.reshape(6)
    fx.run(_inc(6))
    fx.reshape(8)
    fx.run(_inc(8))
    assert rt.hip_graph_capture_count() == 2
    fx.reshape(4)
    fx.run(_inc(4))
    assert rt.hip_graph_capture_count() == 3 and rt.hip_graph_cache_size() == 2
    fx.reshape(8)
    fx.run(_inc(8))
    assert rt.hip_graph_capture_count() == 3
    fx.reshape(6)
    fx.run(_inc(6))
    assert rt.hip_graph_capture_count() == 4 and rt.hip_graph_cache_size() == 2


def test_hipgraph_caches_multiple_graphs_and_drops_dead_ones(B):
    """test_cudagraph.cc:213-232: two graphs on one runtime; a destroyed graph's capture leaves the cache."""
    import gc

    rt = B.RocmRuntime(0)
    first, second = _GraphFixture(B, rt, 2), _GraphFixture(B, rt, 3)
    first.run(_inc(2))
    second.run(_inc(3))
    first.run(_inc(2, 10))
    second.run(_inc(3, 20))
    assert rt.hip_graph_capture_count() == 2 and rt.hip_graph_cache_size() == 2
    del first
    gc.collect()
    second.run(_inc(3, 30))  # the sweep of expired owners happens on the next run (and on invalidation)
    assert rt.hip_graph_cache_size() == 1 and rt.hip_graph_capture_count() == 2
    assert np.array_equal(second.out(), _inc(3, 30))


def test_hipgraph_serializes_threads_on_one_runtime(B):
    """test_cudagraph.cc:234-258: two threads hammer two graphs of ONE runtime: runs serialise on the execution mutex."""
    import threading

    rt = B.RocmRuntime(0)
    first, second = _GraphFixture(B, rt, 2), _GraphFixture(B, rt, 3)
    put(first.input, _inc(2).reshape(2, 2))
    put(second.input, _inc(3).reshape(3, 2))
    errors = []

    def worker(fx):
        try:
            for _ in range(20):
                fx.h.run_with_hipgraph()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=worker, args=(fx,)) for fx in (first, second)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert rt.hip_graph_capture_count() == 2 and rt.hip_graph_cache_size() == 2
    assert np.array_equal(first.out(), _inc(2)) and np.array_equal(second.out(), _inc(3))


def test_hipgraph_independent_runtime_streams(B):
    """test_cudagraph.cc:260-279: two runtimes capture and replay independently; one outlives the other."""
    import gc

    rt2 = B.RocmRuntime(0)
    second = _GraphFixture(B, rt2, 3)
    rt1 = B.RocmRuntime(0)
    first = _GraphFixture(B, rt1, 2)
    first.run(_inc(2))
    second.run(_inc(3))
    first.run(_inc(2, 10))
    second.run(_inc(3, 20))
    assert rt1.hip_graph_capture_count() == 1 and np.array_equal(first.out(), _inc(2, 10))
    del first, rt1
    gc.collect()
    second.run(_inc(3, 30))
    assert rt2.hip_graph_capture_count() == 1 and np.array_equal(second.out(), _inc(3, 30))


def test_hipgraph_two_runtimes_run_concurrently(B):
    """test_cudagraph.cc:260-279 under threads: each runtime has its own stream and ThreadLocal capture mode, so two
    threads may capture / replay at the same time."""
    import threading

    fxs = [_GraphFixture(B, B.RocmRuntime(0), n) for n in (2, 3)]
    errors = []

    def worker(fx, n):
        try:
            for i in range(20):
                fx.run(_inc(n, i))
                assert np.array_equal(fx.out(), _inc(n, i))
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=worker, args=(fx, n)) for fx, n in zip(fxs, (2, 3))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert all(fx.rt.hip_graph_capture_count() == 1 for fx in fxs)


def test_hipgraph_recovers_after_capture_failure(B):
    """test_cudagraph.cc:281-304: a graph whose operator cannot be launched (the reference registers a kernel that
    synchronises inside capture; here InstanceNormalization, which has no Device::ROCM kernel, throws inside the capture) fails loudly,
    leaves the cache and the stream usable, and an earlier capture still replays."""
    rt = B.RocmRuntime(0)
    valid = _GraphFixture(B, rt, 2)
    valid.run(_inc(2))
    h = B.GraphHandler(rt)
    x = h.tensor([2, 3, 4, 4], F32)
    h.instanceNormalization(x, None, h.tensor([3], F32), h.tensor([3], F32), 1e-5)
    h.data_malloc()
    put(x, np.ones((2, 3, 4, 4), np.float32))
    with pytest.raises(RuntimeError):
        h.run_with_hipgraph()
    assert rt.hip_graph_capture_count() == 1 and rt.hip_graph_cache_size() == 1
    valid.run(_inc(2, 5))
    assert rt.hip_graph_capture_count() == 1 and np.array_equal(valid.out(), _inc(2, 5))
    # and a fresh capture on the rebuilt stream works too
    other = _GraphFixture(B, rt, 5)
    other.run(_inc(5, 1))
    assert rt.hip_graph_capture_count() == 2 and np.array_equal(other.out(), _inc(5, 1))


def test_hipgraph_cache_can_be_cleared(B):
    """test_cudagraph.cc:306-320."""
    with pytest.raises(RuntimeError):
        B.RocmRuntime(0, 0)
    rt = B.RocmRuntime(0)
    fx = _GraphFixture(B, rt, 2)
    fx.run(_inc(2))
    rt.clear_hip_graph_cache()
    assert rt.hip_graph_cache_size() == 0 and rt.hip_graph_capture_count() == 1
    fx.run(_inc(2))
    assert rt.hip_graph_capture_count() == 2


def test_replay_survives_workspace_growth_by_a_later_graph(B):
    """A small captured graph whose kernels use scratch (conv_s1 re-packs weights into the workspace) must stay
    replayable after a LARGER graph on the same runtime made the workspace grow: the outgrown block is retired, not
    freed (csrc/runtime.hip), so the first graph's kernel nodes still address live memory."""
    rt = B.RocmRuntime(0)
    rng = np.random.default_rng(7)

    def conv_graph(c, f, hw):
        x = rng.standard_normal((2, c, hw, hw)).astype(np.float16)
        w = (rng.standard_normal((f, c, 3, 3)) / np.sqrt(9 * c)).astype(np.float16)
        h, out = build(B, rt, lambda hd, t: hd.conv(t[0], t[1], None, 1, 1, 1, 1, 1, 1), [(x.shape, F16, x), (w.shape, F16, w)])
        want = R.conv2d(x.astype(np.float64), w.astype(np.float64), 1, 1, 1, 1, 1, 1)
        return h, out, want

    h1, o1, w1 = conv_graph(32, 48, 12)
    h1.run_with_hipgraph()  # capture #1; the workspace grows INSIDE this capture (no eager sizing run)
    assert rt.hip_graph_capture_count() == 1
    assert np.allclose(get(o1).astype(np.float64).reshape(w1.shape), w1, rtol=3e-3, atol=3e-3)
    h2, o2, w2 = conv_graph(256, 512, 14)  # 2.4 MB of re-packed weights: outgrows the first block
    h2.run_with_hipgraph()
    assert np.allclose(get(o2).astype(np.float64).reshape(w2.shape), w2, rtol=3e-3, atol=3e-3)
    h2.run()  # eager too
    for _ in range(3):  # the first graph replays against the retired block
        o1_before = get(o1).copy()
        h1.run_with_hipgraph()
        assert np.array_equal(get(o1), o1_before)
    assert rt.hip_graph_capture_count() == 2
    assert np.allclose(get(o1).astype(np.float64).reshape(w1.shape), w1, rtol=3e-3, atol=3e-3)


def test_conv_weights_are_packed_once_and_repacked_after_copyin(B):
    """Graph weights (no producing operator) are re-packed once: after the first run the packed image is cached, a
    hipGraph capture holds no pack kernel, and copying new weights in drops the image AND the captures that read it
    (the replay after the copy-in re-captures and sees the new weights; the reference's cuDNN path has no weight
    transform to go stale, src/kernels/cuda/conv.cc:143-168)."""
    rt = B.RocmRuntime(0)
    rng = np.random.default_rng(17)
    x = rng.standard_normal((2, 64, 12, 12)).astype(np.float16)
    w1 = (rng.standard_normal((128, 64, 3, 3)) / 24).astype(np.float16)
    w2 = (rng.standard_normal((128, 64, 3, 3)) / 24).astype(np.float16)
    h = B.GraphHandler(rt)
    tx, tw = h.tensor([2, 64, 12, 12], F16), h.tensor([128, 64, 3, 3], F16)
    out = h.relu(h.conv(tx, tw, None, 1, 1, 1, 1, 1, 1), None)
    h.data_malloc()
    put(tx, x)
    put(tw, w1)
    want1 = np.maximum(R.conv2d(x.astype(np.float64), w1.astype(np.float64), 1, 1, 1, 1, 1, 1), 0)
    want2 = np.maximum(R.conv2d(x.astype(np.float64), w2.astype(np.float64), 1, 1, 1, 1, 1, 1), 0)
    h.run_with_hipgraph()
    assert np.allclose(get(out).astype(np.float64).reshape(want1.shape), want1, rtol=3e-3, atol=3e-3)
    put(tx, x)  # activations are re-fed before every run (the planner may place the output on the dead input): no recapture
    h.run_with_hipgraph()
    assert rt.hip_graph_capture_count() == 1
    assert np.allclose(get(out).astype(np.float64).reshape(want1.shape), want1, rtol=3e-3, atol=3e-3)
    put(tx, x)
    put(tw, w2)  # weights: the packed image is stale -> dropped together with the capture
    h.run_with_hipgraph()
    assert rt.hip_graph_capture_count() == 2
    assert np.allclose(get(out).astype(np.float64).reshape(want2.shape), want2, rtol=3e-3, atol=3e-3)
    put(tx, x)
    h.run()
    assert np.allclose(get(out).astype(np.float64).reshape(want2.shape), want2, rtol=3e-3, atol=3e-3)


@pytest.mark.parametrize("code,npdt,tol", [(F32, np.float32, 1e-5), (F16, np.float16, 2e-3)])
def test_lrn_through_reference_executor(B, rocm, code, npdt, tol):
    """h.lrn on Device::ROCM (ONNX LRN semantics; the reference ships the operator, the ONNX import / export of its four
    attributes — pyinfinitensor/tests/test_onnxstub.py:407-428 — and a Cambricon kernel only) vs the oracle, AlexNet-style
    attributes and an even window."""
    rng = np.random.default_rng(9)
    x = (rng.standard_normal((2, 11, 5, 7)) * 3).astype(npdt)
    for size, alpha, beta, bias in ((5, 1e-4, 0.75, 2.0), (4, 0.5, 0.5, 1.0), (1, 2.0, 1.0, 0.5), (31, 1e-2, 0.75, 1.0)):
        h, out = build(B, rocm, lambda hd, t: hd.lrn(t[0], None, alpha, beta, bias, size), [(x.shape, code, x)])
        h.run()
        want = R.lrn(x.astype(np.float64), size, alpha, beta, bias)
        assert np.allclose(get(out).astype(np.float64).reshape(want.shape), want, rtol=tol, atol=tol), (size, alpha)


# ---- the chains in the form the ONNX front-end emits them (pyinfinitensor/onnx.py) -----------------------------------------
def _on_off(B, rocm, fn, ins, hipgraph=False):
    """Runs the graph with launch planning on and off; returns ({on: output as fp64}, {on: fused launches}, plan lines)."""
    got, fused, plan = {}, {}, None
    if PLAN_ONLY:
        h, _ = build(B, B.cpu_runtime(), fn, ins)
        print("\n".join(h.rocm_fusion_plan()))
        pytest.skip("plan only")
    try:
        for on in (True, False):
            rocm.set_fusion(on)
            h, out = build(B, rocm, fn, ins)
            if on:
                plan = h.rocm_fusion_plan()
            before = rocm.fused_launch_count()
            if hipgraph:
                h.run_with_hipgraph()
            else:
                h.run()
            fused[on] = rocm.fused_launch_count() - before
            got[on] = get(out).astype(np.float64)
    finally:
        rocm.set_fusion(True)
    return got, fused, plan


def _weights(t):
    t[0].set_input()  # the graph input: never recycled by the memory planner (graph.cc dataMalloc)
    for x in t[1:]:   # everything else is an initializer of the ONNX graph
        x.set_weight()


@pytest.mark.parametrize("code,npdt,tol", [(F32, np.float32, 1e-4), (F16, np.float16, 4e-3)])
def test_front_end_conv_bias_reshape_chain_is_one_launch(B, rocm, code, npdt, tol):
    """onnx.py:159-190 lowers a Conv with bias to conv -> reshape(bias, [1, F, 1, 1]) -> add; the operator order is
    [Conv, Reshape, Add, Relu]. Planned as ONE launch (the Reshape of a weight is not launched: the epilogue reads the
    weight), equal to the four-kernel run and to the oracle."""
    rng = np.random.default_rng(41)
    c, f = 32, 48
    ins = [((2, c, 10, 12), code, rng.standard_normal((2, c, 10, 12)).astype(npdt)),
           ((f, c, 3, 3), code, (rng.standard_normal((f, c, 3, 3)) / 17).astype(npdt)),
           ((f,), code, rng.standard_normal((f,)).astype(npdt))]

    def fn(h, t):
        _weights(t)
        y = h.conv(t[0], t[1], None, 1, 1, 1, 1, 1, 1)
        return h.relu(h.add(y, h.reshape(t[2], None, [1, f, 1, 1]), None), None)

    got, fused, plan = _on_off(B, rocm, fn, ins)
    assert fused == {True: 1, False: 0} and plan == ["3 conv+bias+relu [0,1,2,3]"], (fused, plan)
    x, w, b = (a.astype(np.float64) for _, _, a in ins)
    want = np.maximum(R.conv2d(x, w, 1, 1, 1, 1, 1, 1) + b.reshape(1, f, 1, 1), 0).reshape(got[True].shape)
    if code == F32:
        assert np.array_equal(got[True], got[False])
    for on in (True, False):
        assert np.allclose(got[on], want, rtol=tol, atol=tol), (on, np.abs(got[on] - want).max())


@pytest.mark.parametrize("bias_first", [True, False])
@pytest.mark.parametrize("tail", ["none", "gelu", "gelu5", "headsplit", "reshape"])
def test_front_end_matmul_add_bias_chains(B, rocm, bias_first, tail):
    """onnx.py:280-290 imports MatMul with no bias: a linear layer is MatMul -> Add(bias) (the exporter puts the bias first).
    The Add folds into the GEMM epilogue, and with it whatever follows: Gelu (one operator or the five-operator opset < 20
    form), the Reshape -> Transpose head split, a Reshape. One launch each; within f16 rounding of the per-operator run and
    of the fp64 oracle (the fused GEMM rounds once where the chain rounds after every operator)."""
    rng = np.random.default_rng(42)
    Bt, S, NH, D, K = 2, 128, 2, 64, 192
    N = NH * D
    ins = [((Bt, S, K), F16, rng.standard_normal((Bt, S, K)).astype(np.float16)),
           ((K, N), F16, (rng.standard_normal((K, N)) / 8).astype(np.float16)),
           ((N,), F16, rng.standard_normal((N,)).astype(np.float16)),
           ((1,), F16, np.array([np.sqrt(2.0)], np.float16)), ((1,), F16, np.array([1.0], np.float16)),
           ((1,), F16, np.array([0.5], np.float16))]

    def fn(h, t):
        _weights(t)
        mm = h.matmul(t[0], t[1], None, False, False, None, B.ActType.Linear, "default")
        y = h.add(t[2], mm, None) if bias_first else h.add(mm, t[2], None)
        if tail == "gelu":
            y = h.gelu(y, None)
        elif tail == "gelu5":
            y = h.mul(h.mul(y, h.add(h.erf(h.div(y, t[3], None), None), t[4], None), None), t[5], None)
        elif tail == "headsplit":
            y = h.transpose(h.reshape(y, None, [Bt, S, NH, D]), None, [0, 2, 1, 3])
        elif tail == "reshape":
            y = h.reshape(y, None, [Bt * S, N])
        return y

    got, fused, plan = _on_off(B, rocm, fn, ins)
    assert fused == {True: 1, False: 0}, (fused, plan)
    a, w, b = (x.astype(np.float64) for _, _, x in ins[:3])
    want = a @ w + b
    if tail.startswith("gelu"):
        want = R.unary("gelu", want)
    elif tail == "headsplit":
        want = want.reshape(Bt, S, NH, D).transpose(0, 2, 1, 3)
    want = want.reshape(got[True].shape)
    for on in (True, False):
        assert np.allclose(got[on], want, rtol=4e-3, atol=6e-3), (on, np.abs(got[on] - want).max())


@pytest.mark.parametrize("kt", ["transpose", "merged"])
@pytest.mark.parametrize("hipgraph", [False, True])
def test_front_end_attention_block_in_exporter_order(B, rocm, kt, hipgraph):
    """A whole self-attention block as a torch export orders it: q = MatMul + Add; k = MatMul, Add, Reshape, Transpose; v
    likewise; THEN q's Reshape + Transpose; Transpose(K) (or the merged Transpose(0, 2, 3, 1) onnxsim leaves); MatMul, Div,
    Add(mask), Softmax, MatMul, Transpose, Reshape; the output projection MatMul + Add; Add(residual) -> LayerNorm. The plan
    is a handful of launches (q / k / v grouped when the layout allows), every operator inside one."""
    rng = np.random.default_rng(43)
    Bt, S, NH, D = 2, 128, 2, 64
    H = NH * D
    mask = np.where(rng.random((Bt, 1, 1, S)) < 0.85, 0.0, -10000.0).astype(np.float16)
    W = lambda: (rng.standard_normal((H, H)) / 11).astype(np.float16)
    bv = lambda: rng.standard_normal((H,)).astype(np.float16)
    ins = [((Bt, S, H), F16, rng.standard_normal((Bt, S, H)).astype(np.float16))]
    for _ in range(4):
        ins += [((H, H), F16, W()), ((H,), F16, bv())]
    ins += [((1,), F16, np.array([np.sqrt(D)], np.float16)), ((Bt, 1, 1, S), F16, mask),
            ((H,), F16, (1 + 0.1 * rng.standard_normal(H)).astype(np.float16)), ((H,), F16, (0.1 * rng.standard_normal(H)).astype(np.float16))]

    def fn(h, t):
        _weights(t)
        lin = B.ActType.Linear
        x0 = h.relu(t[0], None)  # an intermediate, like a previous layer's output
        linear = lambda a, i: h.add(t[2 + 2 * i], h.matmul(a, t[1 + 2 * i], None, False, False, None, lin, "default"), None)
        hd = lambda y: h.transpose(h.reshape(y, None, [Bt, S, NH, D]), None, [0, 2, 1, 3])
        ql = linear(x0, 0)
        kl = linear(x0, 1)
        if kt == "merged":
            kx = h.transpose(h.reshape(kl, None, [Bt, S, NH, D]), None, [0, 2, 3, 1])
        else:
            k = hd(kl)
        v = hd(linear(x0, 2))
        q = hd(ql)
        if kt == "transpose":
            kx = h.transpose(k, None, [0, 1, 3, 2])
        s = h.add(h.div(h.matmul(q, kx, None, False, False, None, lin, "default"), t[9], None), t[10], None)
        ctx = h.matmul(h.softmax(s, None, 3), v, None, False, False, None, lin, "default")
        ctx = h.reshape(h.transpose(ctx, None, [0, 2, 1, 3]), None, [Bt, S, H])
        return h.layerNormalization(h.add(linear(ctx, 3), x0, None), t[11], None, t[12], 1e-5, 2, 1)

    got, fused, plan = _on_off(B, rocm, fn, ins, hipgraph)
    alone = [p for p in plan if " op [" in p]
    # alone: the leading Relu; and the output projection's MatMul when the planner put its bias Add's output on the (dead)
    # attention result — the bias then joins the residual Add + LayerNorm launch instead of the GEMM epilogue
    assert fused[False] == 0 and 3 <= fused[True] <= 7 and len(alone) <= 2, (fused, plan)
    assert any("attention" in p for p in plan) and any("headsplit" in p for p in plan), plan
    f = lambda i: ins[i][2].astype(np.float64)
    x0 = np.maximum(f(0), 0)
    hd = lambda y: y.reshape(Bt, S, NH, D).transpose(0, 2, 1, 3)
    q, k, v = (hd(x0 @ f(1 + 2 * i) + f(2 + 2 * i)) for i in range(3))
    ctx = R.attention(q, k, v, 1.0 / float(np.float16(np.sqrt(D))), mask.astype(np.float64)).transpose(0, 2, 1, 3).reshape(Bt, S, H)
    want = R.layer_norm(ctx @ f(7) + f(8) + x0, f(11), f(12), 1e-5, 2)
    for on in (True, False):
        assert np.allclose(got[on].reshape(want.shape), want, rtol=1e-2, atol=2e-2), (on, np.abs(got[on].reshape(want.shape) - want).max())


@pytest.mark.parametrize("code,npdt,tol", [(F32, np.float32, 1e-4), (F16, np.float16, 6e-3)])
@pytest.mark.parametrize("form", ["ln", "add+ln", "bias+add+ln", "pre-ln"])
def test_decomposed_layer_norm_is_one_launch(B, rocm, code, npdt, tol, form):
    """An opset < 17 export has no LayerNormalization: ReduceMean, Sub, Pow, ReduceMean, Add(eps), Sqrt, Div, Mul(gamma),
    Add(beta) (onnx.py:837,510,528,504,604,522,516). Nine operators planned as one layer_norm launch — alone, behind the
    residual Add (add_norm), behind MatMul-bias Add + residual Add (bias_add_norm), and in pre-LN position where the
    normalised tensor has other readers (it then stays materialised). fp32 math inside; the per-operator run rounds each
    step to the storage type."""
    rng = np.random.default_rng(44)
    Bt, S, H = 2, 96, 256
    ins = [((Bt, S, H), code, (rng.standard_normal((Bt, S, H)) * 2 + 0.5).astype(npdt)),
           ((H,), code, (1 + 0.1 * rng.standard_normal(H)).astype(npdt)), ((H,), code, (0.1 * rng.standard_normal(H)).astype(npdt)),
           ((1,), code, np.array([2.0], npdt)), ((1,), code, np.array([1e-5], npdt)),
           ((Bt, S, H), code, rng.standard_normal((Bt, S, H)).astype(npdt)), ((H,), code, rng.standard_normal((H,)).astype(npdt))]

    def fn(h, t):
        for i in (1, 2, 3, 4, 6):
            t[i].set_weight()
        x = h.relu(t[0], None)
        if form == "add+ln":
            x = h.add(x, t[5], None)
        elif form == "bias+add+ln":
            x = h.add(h.add(t[6], x, None), t[5], None)
        d = h.sub(x, h.reduceMean(x, None, [2], True), None)
        var = h.reduceMean(h.pow(d, t[3], None), None, [2], True)
        y = h.add(h.mul(h.div(d, h.sqrt(h.add(var, t[4], None), None), None), t[1], None), t[2], None)
        return h.add(y, x, None) if form == "pre-ln" else y

    got, fused, plan = _on_off(B, rocm, fn, ins)
    assert fused[False] == 0 and fused[True] == 1 and sum("layer" in p for p in plan) == 1, (fused, plan)
    f = lambda i: ins[i][2].astype(np.float64)
    x = np.maximum(f(0), 0)
    if form == "add+ln":
        x = x + f(5)
    elif form == "bias+add+ln":
        x = x + f(6) + f(5)
    want = R.layer_norm(x, f(1), f(2), float(ins[4][2][0]), 2)
    if form == "pre-ln":
        want = want + x
    for on in (True, False):
        assert np.allclose(got[on].reshape(want.shape), want, rtol=tol, atol=tol * 4), (on, np.abs(got[on].reshape(want.shape) - want).max())


def test_plan_follows_a_constant_the_host_overwrites(B, rocm):
    """The planner reads one-element constants back to recognise Pow(d, 2) / sqrt 2 / 0.5. Writing a different value over
    such a constant must re-plan — also under run_with_hipgraph, whose captured launches embed the old decision."""
    rng = np.random.default_rng(45)
    Bt, S, H = 2, 64, 128
    x = (rng.standard_normal((Bt, S, H)) + 0.3).astype(np.float32)
    h = B.GraphHandler(rocm)
    tx = h.tensor([Bt, S, H], F32)
    tx.set_input()
    tg, tb, two, eps = (h.tensor(list(s), F32) for s in ((H,), (H,), (1,), (1,)))
    for t in (tg, tb, two, eps):
        t.set_weight()
    d = h.sub(tx, h.reduceMean(tx, None, [2], True), None)
    var = h.reduceMean(h.pow(d, two, None), None, [2], True)
    out = h.add(h.mul(h.div(d, h.sqrt(h.add(var, eps, None), None), None), tg, None), tb, None)
    h.data_malloc()
    g, b = (1 + 0.1 * rng.standard_normal(H)).astype(np.float32), (0.1 * rng.standard_normal(H)).astype(np.float32)
    for t, a in ((tx, x), (tg, g), (tb, b), (two, np.array([2.0], np.float32)), (eps, np.array([1e-5], np.float32))):
        put(t, a)

    def want(p):
        X = x.astype(np.float64)
        dd = X - X.mean(-1, keepdims=True)
        return dd / np.sqrt((np.abs(dd) ** p).mean(-1, keepdims=True) + 1e-5) * g + b

    for run in (h.run, h.run_with_hipgraph):
        put(two, np.array([2.0], np.float32))
        run()
        assert any("layer_norm(decomposed)" in p for p in h.rocm_fusion_plan())
        assert np.allclose(get(out, (Bt, S, H)), want(2), rtol=1e-4, atol=1e-4)
        put(two, np.array([4.0], np.float32))  # no longer a variance: the nine operators must run as written
        run()
        assert not any("layer_norm" in p for p in h.rocm_fusion_plan())
        assert np.allclose(get(out, (Bt, S, H)), want(4), rtol=1e-3, atol=1e-3)


def test_sunk_chain_is_cut_when_its_input_is_recycled(B, rocm):
    """q = MatMul + Add is issued first, reshaped LAST; in between, operators recycle the storage of the MatMul's (dead)
    input. The planned launch would read that input at the Reshape's position — the planner must notice and cut the chain
    there. Whatever it decides, the result equals the per-operator run."""
    rng = np.random.default_rng(46)
    Bt, S, NH, D = 2, 128, 2, 64
    H = NH * D
    ins = [((Bt, S, H), F16, rng.standard_normal((Bt, S, H)).astype(np.float16)),
           ((H, H), F16, (rng.standard_normal((H, H)) / 11).astype(np.float16)), ((H,), F16, rng.standard_normal((H,)).astype(np.float16))]

    def fn(h, t):
        _weights(t)
        a = h.relu(t[0], None)           # dies after the MatMul
        other = h.tanh(t[0], None)
        ql = h.add(t[2], h.matmul(a, t[1], None, False, False, None, B.ActType.Linear, "default"), None)
        for _ in range(3):               # same-size temporaries: the allocator hands them a's block
            other = h.sigmoid(h.neg(other, None), None)
        q = h.transpose(h.reshape(ql, None, [Bt, S, NH, D]), None, [0, 2, 1, 3])
        return h.add(q, h.transpose(h.reshape(other, None, [Bt, S, NH, D]), None, [0, 2, 1, 3]), None)

    got, fused, plan = _on_off(B, rocm, fn, ins)
    assert np.isfinite(got[True]).all()
    assert np.allclose(got[True], got[False], rtol=4e-3, atol=6e-3), (np.abs(got[True] - got[False]).max(), plan)


@pytest.mark.parametrize("form", ["matmul", "conv"])
def test_output_forwarding_when_the_planned_buffer_sits_on_an_operand(B, rocm, form):
    """The memory planner recycles a producer's dead input for the output of the bias Add behind it — exactly the buffer a
    fused GEMM / conv must not write while it still reads that input. The plan then writes the fused result into the
    producer's OWN output buffer and forwards every reader of the final tensor there (ForwardMap): still one launch, no
    copy; equal to the per-operator run."""
    rng = np.random.default_rng(47)
    if form == "matmul":
        m, k = 512, 256
        ins = [((m, k), F16, rng.standard_normal((m, k)).astype(np.float16)),
               ((k, k), F16, (rng.standard_normal((k, k)) / 16).astype(np.float16)), ((k,), F16, rng.standard_normal((k,)).astype(np.float16))]

        def fn(h, t):
            _weights(t)
            a = h.relu(t[0], None)  # an intermediate that dies at the MatMul: the bias Add's output (same size) lands on it
            y = h.add(t[2], h.matmul(a, t[1], None, False, False, None, B.ActType.Linear, "default"), None)
            return h.abs(h.sigmoid(y, None), None)
    else:
        c = 64
        ins = [((4, c, 16, 16), F16, rng.standard_normal((4, c, 16, 16)).astype(np.float16)),
               ((c, c, 1, 1), F16, (rng.standard_normal((c, c, 1, 1)) / 8).astype(np.float16)), ((c,), F16, rng.standard_normal((c,)).astype(np.float16))]

        def fn(h, t):
            _weights(t)
            a = h.relu(t[0], None)
            y = h.relu(h.add(h.conv(a, t[1], None, 0, 0, 1, 1, 1, 1), h.reshape(t[2], None, [1, c, 1, 1]), None), None)
            return h.abs(h.sigmoid(y, None), None)

    f0 = 0 if PLAN_ONLY else rocm.forwarded_output_count()
    got, fused, plan = _on_off(B, rocm, fn, ins)
    assert np.allclose(got[True], got[False], rtol=4e-3, atol=6e-3), (np.abs(got[True] - got[False]).max(), plan)
    fwd = [p for p in plan if "forwarded" in p]
    if fwd:  # (the layout decides; when it happened, it must have been counted)
        assert rocm.forwarded_output_count() > f0
    x, w, b = (a.astype(np.float64) for _, _, a in ins)
    r = np.maximum(x, 0)
    want = np.abs(1 / (1 + np.exp(-(r @ w + b)))) if form == "matmul" else None
    if want is not None:
        assert np.allclose(got[True].reshape(want.shape), want, rtol=4e-3, atol=4e-3)


def test_forwarding_declined_when_another_branch_owns_the_convs_block(B, rocm):
    """tests/_fwd_graph.py on the device: another branch's Tanh was planned onto the Conv operator's own output block and is
    read after the chain's slot (round-3 advisor finding). With forwarding from the chain's slot only, the fused kernel
    clobbered it; the result must equal the oracle and the per-operator run."""
    import _fwd_graph as G

    want = G.oracle(G.build(B, B.cpu_runtime())[2])
    res = {}
    for on in (True, False):
        rocm.set_fusion(on)
        try:
            h, t, feeds = G.build(B, rocm)
            for k, a in feeds.items():
                put(t[k], a)
            assert not any("forwarded" in p for p in h.rocm_fusion_plan()) or not on
            h.run()
            res[on] = get(t["f2"]).astype(np.float64)
        finally:
            rocm.set_fusion(True)
    assert np.allclose(res[True], want, rtol=4e-3, atol=4e-3), np.abs(res[True] - want).max()
    assert np.allclose(res[True], res[False], rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("code,npdt", [(F16, np.float16)])
def test_stem_chain_is_one_launch(B, rocm, code, npdt):
    """Conv(7 x 7 / 2) -> Reshape(bias) -> Add -> Relu -> MaxPool(3 x 3 / 2 / 1) — ResNet's stem in the front-end's form — planned
    as ONE launch (conv_stem.hip: the conv tile is pooled out of LDS) and equal to the per-operator run and the oracle."""
    rng = np.random.default_rng(61)
    x = (rng.random((2, 3, 64, 64)) * 2 - 0.5).astype(npdt)
    w = (rng.standard_normal((64, 3, 7, 7)) * np.sqrt(2 / 147)).astype(npdt)
    b = (rng.standard_normal(64) * 0.2).astype(npdt)
    res = {}
    for on in (True, False):
        rocm.set_fusion(on)
        try:
            h = B.GraphHandler(rocm)
            tx, tw, tb = h.tensor([2, 3, 64, 64], code), h.tensor([64, 3, 7, 7], code), h.tensor([64], code)
            tx.set_input()
            tw.set_weight()
            tb.set_weight()
            y = h.relu(h.add(h.conv(tx, tw, None, 3, 3, 2, 2, 1, 1), h.reshape(tb, None, [1, 64, 1, 1]), None), None)
            out = h.relu(h.maxPool(y, None, 3, 3, 1, 1, 1, 1, 2, 2, 0), None)  # (a consumer behind the pool)
            h.data_malloc()
            for t_, a_ in ((tx, x), (tw, w), (tb, b)):
                put(t_, a_)
            if on:
                plan = h.rocm_fusion_plan()
                assert any("conv+bias+relu+maxpool (stem)" in p for p in plan), plan
            h.run()
            res[on] = get(out).astype(np.float64)
        finally:
            rocm.set_fusion(True)
    conv = R.conv2d(x.astype(np.float64), w.astype(np.float64), 3, 3, 2, 2, 1, 1) + b.astype(np.float64).reshape(1, 64, 1, 1)
    want = R.pool2d(np.maximum(conv.astype(npdt).astype(np.float64), 0), "max", 3, 3, 1, 1, 1, 1, 2, 2, 0)
    assert np.allclose(res[True].reshape(want.shape), want, rtol=3e-3, atol=3e-3), np.abs(res[True].reshape(want.shape) - want).max()
    assert np.allclose(res[True], res[False], rtol=3e-3, atol=3e-3)


@pytest.mark.parametrize("ct", ["bf16", "fp16", "tf32", "default"])
def test_matmul_compute_type_through_the_reference_operator(B, rocm, ct):
    """MatmulObj carries `computeType` (operators/matmul.h:67; the front-end passes `matmul_compute_type`, onnx.py:41-47): the
    plugin honours "bf16" / "fp16" for fp32 graphs (16-bit products, fp32 sums and output) and multiplies exactly for
    "default" / "tf32"."""
    rng = np.random.default_rng(71)
    a = rng.standard_normal((256, 512)).astype(np.float32)
    w = (rng.standard_normal((512, 384)) / 22).astype(np.float32)
    h = B.GraphHandler(rocm)
    ta, tw = h.tensor([256, 512], F32), h.tensor([512, 384], F32)
    ta.set_input()
    tw.set_weight()
    out = h.matmul(ta, tw, None, False, False, None, B.ActType.Linear, ct)
    h.data_malloc()
    put(ta, a)
    put(tw, w)
    h.run()
    got = get(out).astype(np.float64)
    exact = a.astype(np.float64) @ w.astype(np.float64)
    if ct in ("bf16", "fp16"):
        name = {"bf16": "bf16", "fp16": "f16"}[ct]
        want = R.matmul(R.round_to(a, name), R.round_to(w, name))
        assert np.allclose(got, want, rtol=2e-5, atol=1e-4), np.abs(got - want).max()
        assert np.abs(got - exact).max() > 1e-5  # the attribute took effect
    else:
        assert np.allclose(got, exact, rtol=1e-4, atol=2e-5), np.abs(got - exact).max()
