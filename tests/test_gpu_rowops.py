"""Softmax / LayerNorm / RMSNorm parity on a real MI355X vs the oracle and the reference KATs."""
import numpy as np
import pytest
import torch
from conftest import kat

from infinitensor_amd import ops
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu
SM = "test/kernels/cuda/test_cuda_softmax.cc"
LN = "test/kernels/cuda/test_cuda_layernorm.cc"
TD = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
TOL = {"f32": (1e-4, 1e-6), "f16": (2e-3, 1e-3), "bf16": (1.6e-2, 8e-3)}  # (rtol, atol): fp32 gate / 1 ulp of storage


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t.to(dtype) if dtype is not None else t).cuda()


def host(t):
    return t.float().cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("axis,in_line,out_line", [(0, 70, 73), (1, 83, 86), (2, 96, 99), (3, 107, 110)])
def test_softmax_reference_kats_fp32(rt, axis, in_line, out_line):
    x = kat(SM, in_line, "float").astype(np.float32).reshape(2, 3, 2, 2)
    y = ops.softmax(rt, dev(x), axis)
    assert R.equal_data(host(y).ravel(), kat(SM, out_line, "float"), 1e-6)


@pytest.mark.parametrize("axis,line", [(0, 119), (1, 126)])
def test_softmax_reference_kats_fp16(rt, axis, line):
    x = R.value((2, 3, 2, 2), 2.0, np.float16)
    y = ops.softmax(rt, dev(x), axis)
    assert R.equal_data(host(y).ravel(), kat(SM, line, "float"), 1e-3)


SOFTMAX_SHAPES = [
    ((4, 12, 64, 512), 3), ((7, 1000), 1), ((3, 5, 17), 2), ((6, 33, 40), 1), ((5, 64, 3), 0),
    ((2, 2048), 1), ((3, 5000), 1), ((2, 20000), 1), ((16, 1), 1), ((1, 7), 0), ((4, 513), 1),
    ((3, 8192), 1), ((2, 16384), 1), ((3, 4104), 1), ((300, 6144), 1),  # block-resident rows (softmax_blockreg_kernel)
]


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("shape,axis", SOFTMAX_SHAPES)
def test_softmax_vs_oracle(rt, shape, axis, dt):
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(shape) * 3).astype(np.float32)
    xs = R.round_to(x, dt)
    y = ops.softmax(rt, dev(x, TD[dt]), axis)
    want = R.softmax(xs, axis)
    rtol, atol = TOL[dt]
    got = host(y)
    assert np.allclose(got, want, rtol=rtol, atol=atol)
    assert np.allclose(got.sum(axis=axis), 1.0, atol=5e-2 if dt != "f32" else 1e-5)  # rows sum to 1


def test_softmax_extreme_values(rt):
    x = np.array([[-1e4, 0, 1e4], [88.0, 88.5, 89.0], [-np.inf, 0.0, 1.0]], dtype=np.float32)
    y = host(ops.softmax(rt, dev(x), 1))
    assert np.allclose(y, R.softmax(x, 1), rtol=1e-4, atol=1e-7)
    assert np.isfinite(y).all()


@pytest.mark.parametrize("xi,si,yi,bi", [(153, 157, 158, 165), (168, 172, 173, 180), (183, 187, 188, 195), (198, 202, 203, None)])
def test_layernorm_reference_kats_fp32(rt, xi, si, yi, bi):
    x = kat(LN, xi, "float").astype(np.float32).reshape(2, 3, 2, 3)
    scale = kat(LN, si, "float").astype(np.float32)
    bias = kat(LN, bi, "float").astype(np.float32) if bi else None
    y = ops.layer_norm(rt, dev(x), dev(scale), dev(bias) if bias is not None else None, 1e-5, 3)
    assert R.equal_data(host(y).ravel(), kat(LN, yi, "float"), 2e-6)


def test_layernorm_reference_kat_fp16(rt):
    two = lambda s: dev(R.value(s, 2, np.float16))
    y = ops.layer_norm(rt, two((2, 3, 2, 3)), two((3,)), two((3,)), 1e-5, 3)
    assert R.equal_data(host(y).ravel(), kat(LN, 216, "float"), 1e-3)


LN_SHAPES = [((32, 512, 768), -1), ((5, 7, 33), -1), ((4, 3, 8, 16), 2), ((9, 1024), 1), ((3, 4096), 1),
             ((2, 10000), 1), ((6, 1), 1), ((4, 257), 1),
             ((5, 8192), 1), ((3, 16384), 1), ((2, 12288), 1), ((300, 5120), 1)]  # block-resident rows (norm_blockreg_kernel)


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("shape,axis", LN_SHAPES)
@pytest.mark.parametrize("with_bias", [True, False])
def test_layernorm_vs_oracle(rt, shape, axis, dt, with_bias):
    if shape[0] == 32 and dt != "f16":
        shape = (4,) + shape[1:]
    rng = np.random.default_rng(9)
    x = (rng.standard_normal(shape) * 2 + 0.5).astype(np.float32)
    nshape = shape[axis:] if axis >= 0 else shape[axis:]
    scale = rng.standard_normal(nshape).astype(np.float32)
    bias = rng.standard_normal(nshape).astype(np.float32) if with_bias else None
    y = ops.layer_norm(rt, dev(x, TD[dt]), dev(scale, TD[dt]), dev(bias, TD[dt]) if with_bias else None, 1e-5, axis)
    want = R.layer_norm(R.round_to(x, dt), R.round_to(scale, dt), R.round_to(bias, dt) if with_bias else None, 1e-5, axis)
    rtol, atol = TOL[dt]
    assert np.allclose(host(y), want, rtol=rtol, atol=max(atol, rtol))


def test_layernorm_scalar_scale_and_large_mean(rt):
    rng = np.random.default_rng(2)
    x = (rng.standard_normal((16, 768)) + 1000.0).astype(np.float32)  # cancellation-prone
    s = np.array([0.3], dtype=np.float32)
    b = rng.standard_normal((768,)).astype(np.float32)
    y = ops.layer_norm(rt, dev(x), dev(s), dev(b), 1e-5, -1)
    assert np.allclose(host(y), R.layer_norm(x, s, b, 1e-5, -1), rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
def test_rmsnorm_vs_oracle(rt, dt):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 37, 4096)).astype(np.float32)
    w = rng.standard_normal((4096,)).astype(np.float32)
    y = ops.rms_norm(rt, dev(x, TD[dt]), dev(w, TD[dt]), 1e-5)
    rtol, atol = TOL[dt]
    assert np.allclose(host(y), R.rms_norm(R.round_to(x, dt), R.round_to(w, dt), 1e-5), rtol=rtol, atol=max(atol, rtol))


def test_empty_inputs_are_noops(rt):
    x = torch.empty((0, 16), device="cuda")
    assert ops.softmax(rt, x, 1).shape == (0, 16)
    assert ops.layer_norm(rt, x, torch.ones(16, device="cuda"), None, 1e-5, 1).shape == (0, 16)


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
def test_rope_vs_oracle(rt, dt):
    rng = np.random.default_rng(31)
    x = rng.standard_normal((2, 37, 512)).astype(np.float32)
    pos = np.tile(np.arange(37, dtype=np.int64) * 3, (2, 1))
    for pdt in (np.int64, np.int32):
        y = ops.rope(rt, dev(pos.astype(pdt)), dev(x, TD[dt]), 128)
        want = R.rope(pos, R.round_to(x, dt), 128)
        rtol, atol = TOL[dt]
        assert np.allclose(host(y), want, rtol=rtol, atol=max(atol, 2e-4))


def test_rope_reference_kat(rt):
    """test_cuda_rope.cc:17-31: once with the row zero-padded to one 128-wide head, once at the reference's exact
    shape {1, 1, 32} (head dim 128 > dim_model: a partial head whose partner columns count as 0)."""
    x = np.zeros((1, 1, 128), np.float32)
    x[..., :32] = 1
    y = ops.rope(rt, dev(np.array([[1]], np.int32)), dev(x), 128)
    assert R.equal_data(host(y)[0, 0, :32], kat("test/kernels/cuda/test_cuda_rope.cc", 29, "float"), 2e-6)
    y = ops.rope(rt, dev(np.array([[1]], np.int32)), dev(np.ones((1, 1, 32), np.float32)), 128)
    assert R.equal_data(host(y)[0, 0], kat("test/kernels/cuda/test_cuda_rope.cc", 29, "float"), 2e-6)


def test_rope_partial_trailing_head(rt):
    """dim_model = 2 full heads + 40 columns: full heads rotate normally, the partial head's columns whose partner
    lies beyond the row keep x * cos."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 9, 2 * 64 + 40)).astype(np.float32)
    pos = np.arange(9, dtype=np.int32)[None]
    y = ops.rope(rt, dev(pos), dev(x), 64)
    assert np.allclose(host(y), R.rope(pos, x, 64), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("shape", [(64, 768), (3, 7, 1024), (5, 33), (2, 4096), (4, 520)])
@pytest.mark.parametrize("rms", [False, True])
def test_add_norm_matches_the_chain(rt, shape, rms, dt):
    """infini_rocm_add_norm vs Norm(Add(a, b)): the sum is rounded like the chain's, so the two agree except where the
    fp32 result sits on a rounding tie of the output type (a handful of elements per 10^5, one output ulp apart)."""
    rng = np.random.default_rng(19)
    a, b = (dev(rng.standard_normal(shape).astype(np.float32), TD[dt]) for _ in range(2))
    g = dev(rng.standard_normal(shape[-1:]).astype(np.float32), TD[dt])
    be = None if rms else dev(rng.standard_normal(shape[-1:]).astype(np.float32), TD[dt])
    fused = ops.add_layer_norm(rt, a, b, g, be, 1e-5, rms)
    s = ops.binary(rt, "add", a, b)
    chain = ops.rms_norm(rt, s, g, 1e-5) if rms else ops.layer_norm(rt, s, g, be, 1e-5, -1)
    ulp = {"f32": 2.0 ** -22, "f16": 2.0 ** -10, "bf16": 2.0 ** -7}[dt]
    f, c = fused.float(), chain.float()
    assert torch.allclose(f, c, rtol=ulp, atol=ulp * 1e-2)
    assert (f != c).float().mean().item() < 1e-3


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("shape", [(64, 768), (3, 7, 1024), (5, 33), (2, 4096), (4, 520)])
@pytest.mark.parametrize("rms", [False, True])
def test_bias_add_norm_matches_the_chain(rt, shape, rms, dt):
    """infini_rocm_bias_add_norm vs Norm(Add(Add(a, row bias), b)) — the chain the ONNX front-end emits behind a linear layer
    (MatMul -> Add(bias) -> Add(residual) -> LayerNormalization): both sums rounded like their own Add kernels."""
    rng = np.random.default_rng(23)
    a, b = (dev(rng.standard_normal(shape).astype(np.float32), TD[dt]) for _ in range(2))
    pre = dev(rng.standard_normal(shape[-1:]).astype(np.float32), TD[dt])
    g = dev(rng.standard_normal(shape[-1:]).astype(np.float32), TD[dt])
    be = None if rms else dev(rng.standard_normal(shape[-1:]).astype(np.float32), TD[dt])
    fused = ops.add_layer_norm(rt, a, b, g, be, 1e-5, rms, pre=pre)
    s = ops.binary(rt, "add", ops.binary(rt, "add", a, pre), b)
    chain = ops.rms_norm(rt, s, g, 1e-5) if rms else ops.layer_norm(rt, s, g, be, 1e-5, -1)
    ulp = {"f32": 2.0 ** -22, "f16": 2.0 ** -10, "bf16": 2.0 ** -7}[dt]
    f, c = fused.float(), chain.float()
    assert torch.allclose(f, c, rtol=ulp, atol=ulp * 1e-2)
    assert (f != c).float().mean().item() < 1e-3


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("shape", [(6, 4096), (5, 33), (3, 8200), (64, 768)])
@pytest.mark.parametrize("alias", ["b", "a"])
@pytest.mark.parametrize("with_pre", [True, False])
def test_bias_add_norm_in_place_over_an_operand(rt, shape, dt, alias, with_pre):
    """The planner lets the norm's output sit on the residual b (or on a): `ops.add_layer_norm(..., out=b, pre=...)` must
    give the out-of-place result bit for bit — on the fused kernel AND on the fallback for rows beyond 4 KiB / unaligned rows
    (round-3 advisor finding: the fallback stored y = a + pre over b and then added the overwritten b)."""
    rng = np.random.default_rng(29)
    a, b = (dev(rng.standard_normal(shape).astype(np.float32), TD[dt]) for _ in range(2))
    pre = dev(rng.standard_normal(shape[-1:]).astype(np.float32), TD[dt]) if with_pre else None
    g = dev(rng.standard_normal(shape[-1:]).astype(np.float32), TD[dt])
    be = dev(rng.standard_normal(shape[-1:]).astype(np.float32), TD[dt])
    want = ops.add_layer_norm(rt, a, b, g, be, 1e-5, False, pre=pre)
    s = ops.binary(rt, "add", ops.binary(rt, "add", a, pre), b) if with_pre else ops.binary(rt, "add", a, b)
    chain = ops.layer_norm(rt, s, g, be, 1e-5, -1)
    ulp = {"f32": 2.0 ** -22, "f16": 2.0 ** -10, "bf16": 2.0 ** -7}[dt]
    assert torch.allclose(want.float(), chain.float(), rtol=ulp, atol=ulp * 1e-2)
    tgt = (b if alias == "b" else a).clone()
    args = (a, tgt) if alias == "b" else (tgt, b)
    got = ops.add_layer_norm(rt, *args, g, be, 1e-5, False, out=tgt, pre=pre)
    rt.sync()
    assert got.data_ptr() == tgt.data_ptr() and torch.equal(got, want)


@pytest.mark.parametrize("dt", ["f16", "f32"])
def test_softmax_bert_full_size_properties(rt, dt):
    """BASELINE config 4's softmax at its full size ([32, 12, 512, 512] = 196 608 rows of 512) through size-independent
    properties: every row sums to 1, is non-negative and keeps the arg-max of its input; softmax(x + c) == softmax(x)
    for a per-row constant; 64 sampled rows equal the oracle."""
    import torch

    tdt = {"f16": torch.float16, "f32": torch.float32}[dt]
    g = torch.Generator(device="cuda").manual_seed(5)
    x = (torch.randn(32, 12, 512, 512, device="cuda", generator=g) * 3).to(tdt)
    y = ops.softmax(rt, x, 3)
    rt.sync()
    sums = y.float().sum(-1)
    tol = 2e-3 if dt == "f16" else 1e-5
    assert float((sums - 1).abs().max()) <= tol * 4
    assert float(y.float().min()) >= 0
    assert bool((y.float().argmax(-1) == x.float().argmax(-1)).float().mean() > 0.999)  # ties in f16 storage aside
    if dt == "f32":
        shift = torch.randn(32, 12, 512, 1, device="cuda", generator=g)
        y2 = ops.softmax(rt, (x + shift).contiguous(), 3)
        assert float((y2 - y).abs().max()) <= 2e-6
    rows = np.random.default_rng(0).integers(0, 32 * 12 * 512, 64)
    xs = x.view(-1, 512)[torch.from_numpy(rows).cuda()].float().cpu().numpy().astype(np.float64)
    ys = y.view(-1, 512)[torch.from_numpy(rows).cuda()].float().cpu().numpy().astype(np.float64)
    assert np.allclose(ys, R.softmax(xs, 1), rtol=tol, atol=tol)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.bfloat16])
def test_rope_head_split_store_is_the_transposed_result(rt, dt):
    """infini_rocm_rope_headsplit: RoPE -> Reshape([B, S, H, D]) -> Transpose(0, 2, 1, 3) as one pass — the same bits as the
    plain RoPE followed by the two movement ops."""
    g = torch.Generator().manual_seed(5)
    B_, S, H, D = 2, 37, 3, 128
    x = torch.randn(B_, S, H * D, generator=g).to(dt).cuda()
    pos = torch.arange(S, dtype=torch.int32).repeat(B_, 1).cuda()
    plain = ops.rope(rt, pos, x, D)
    want = ops.transpose(rt, plain.view(B_, S, H, D), (0, 2, 1, 3))
    got = ops.rope(rt, pos, x, D, head_split=True)
    rt.sync()
    assert tuple(got.shape) == (B_, H, S, D) and torch.equal(got, want)
