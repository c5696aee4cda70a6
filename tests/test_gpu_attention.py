"""Fused prefill attention (csrc/attention.hip) vs the oracle and vs the unfused five-kernel chain it replaces."""
import numpy as np
import pytest
import torch

from infinitensor_amd import ops
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu
TD = {"f16": torch.float16, "bf16": torch.bfloat16}
TOL = {"f16": 3e-3, "bf16": 2e-2}


def dev(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).cuda()


def host(t):
    return t.float().cpu().numpy().astype(np.float64)


CASES = [
    # b, h, sq, sk, d, mask, causal
    (2, 3, 128, 128, 64, True, False),
    (1, 2, 512, 512, 64, True, False),    # BERT head
    (2, 2, 100, 77, 64, True, False),     # ragged: partial query tile, partial key tile
    (1, 4, 256, 256, 128, False, True),   # Llama head, causal
    (1, 2, 65, 65, 128, False, True),
    (1, 1, 40, 200, 64, False, True),     # causal with Sk > Sq (bottom-right aligned)
    (3, 1, 17, 5, 128, True, False),
    (1, 2, 300, 300, 64, False, False),
]


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("case", CASES)
def test_attention_vs_oracle(rt, case, dt):
    b, h, sq, sk, d, use_mask, causal = case
    rng = np.random.default_rng(abs(hash(case)) % 2 ** 32)
    q = rng.standard_normal((b, h, sq, d)).astype(np.float32)
    k = rng.standard_normal((b, h, sk, d)).astype(np.float32)
    v = rng.standard_normal((b, h, sk, d)).astype(np.float32)
    scale = 1.0 / np.sqrt(d)
    mask = None
    if use_mask:  # BERT padding mask: 0 for kept keys, large negative for padded ones (never all padded)
        mask = np.where(rng.random((b, sk)) < 0.8, 0.0, -10000.0).astype(np.float32)
        mask[:, 0] = 0.0
    qd, kd, vd = dev(q, TD[dt]), dev(k, TD[dt]), dev(v, TD[dt])
    md = dev(mask, TD[dt]) if use_mask else None
    y = ops.attention(rt, qd, kd, vd, scale, md, causal)
    want = R.attention(R.round_to(q, dt), R.round_to(k, dt), R.round_to(v, dt), scale,
                       None if mask is None else R.round_to(mask, dt)[:, None, None, :], causal)
    assert np.allclose(host(y), want, rtol=TOL[dt], atol=TOL[dt])


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("scale", [-0.125, 0.0, 3.0])
@pytest.mark.parametrize("d,causal,use_mask", [(64, False, True), (128, True, False), (64, True, True)])
def test_attention_scale_sign_and_zero(rt, dt, scale, d, causal, use_mask):
    """The kernel keeps the scores in q.k units and applies the scale inside the exponent, which needs a positive factor:
    the sign of a negative scale goes into Q, a zero scale zeroes Q (attention.hip). Same result as the oracle's
    softmax(scale * Q K^T + mask) V for every sign, also through a device-memory scale used as a divisor."""
    rng = np.random.default_rng(17)
    b, h, sq, sk = 2, 2, 70, 150
    q = rng.standard_normal((b, h, sq, d)).astype(np.float32)
    k = rng.standard_normal((b, h, sk, d)).astype(np.float32)
    v = rng.standard_normal((b, h, sk, d)).astype(np.float32)
    mask = None
    if use_mask:
        mask = np.where(rng.random((b, sk)) < 0.8, 0.0, -10000.0).astype(np.float32)
        mask[:, 0] = 0.0
    qd, kd, vd = dev(q, TD[dt]), dev(k, TD[dt]), dev(v, TD[dt])
    md = dev(mask, TD[dt]) if use_mask else None
    want = R.attention(R.round_to(q, dt), R.round_to(k, dt), R.round_to(v, dt), scale,
                       None if mask is None else R.round_to(mask, dt)[:, None, None, :], causal)
    y = ops.attention(rt, qd, kd, vd, scale, md, causal)
    assert np.allclose(host(y), want, rtol=TOL[dt], atol=TOL[dt])
    if scale != 0.0:  # the graph's Div(scalar) form: scale = 1 / value held in device memory
        inv = torch.tensor([1.0 / scale], dtype=TD[dt]).cuda()
        eff = 1.0 / float(inv.float().item())  # the 16-bit rounding of the constant is part of the graph
        want_div = R.attention(R.round_to(q, dt), R.round_to(k, dt), R.round_to(v, dt), eff,
                               None if mask is None else R.round_to(mask, dt)[:, None, None, :], causal)
        y2 = ops.attention(rt, qd, kd, vd, inv, md, causal, scale_is_div=True)
        assert np.allclose(host(y2), want_div, rtol=TOL[dt], atol=TOL[dt])


def test_attention_equals_the_unfused_chain(rt):
    """Same graph as BERT emits: MatMul(q, k^T) -> Div(sqrt d) -> Add(mask) -> Softmax -> MatMul(p, v), scale taken
    from device memory as the graph's scalar constant."""
    rng = np.random.default_rng(3)
    b, h, s, d = 2, 4, 192, 64
    q, k, v = (dev(rng.standard_normal((b, h, s, d)).astype(np.float32), torch.float16) for _ in range(3))
    mask = dev(np.where(rng.random((b, 1, 1, s)) < 0.9, 0.0, -10000.0).astype(np.float32), torch.float16)
    sc = torch.tensor([np.sqrt(d)], dtype=torch.float16).cuda()
    sm = ops.matmul(rt, q, k, None, False, True)
    sm = ops.binary(rt, "add", ops.binary(rt, "div", sm, sc), mask)
    chain = ops.matmul(rt, ops.softmax(rt, sm, -1), v)
    fused = ops.attention(rt, q, k, v, sc, mask.reshape(b, s), scale_is_div=True)
    assert np.allclose(host(fused), host(chain), rtol=4e-3, atol=4e-3)


def test_attention_fully_masked_rows_and_errors(rt):
    q = torch.randn(1, 1, 16, 64, device="cuda").half()
    k = torch.randn(1, 1, 16, 64, device="cuda").half()
    y = ops.attention(rt, q, k, k.clone(), 0.125, None, True)
    assert torch.isfinite(y).all()
    with pytest.raises(RuntimeError):
        ops.attention(rt, q[..., :32].contiguous(), k[..., :32].contiguous(), k[..., :32].contiguous(), 1.0)
    with pytest.raises(RuntimeError):
        ops.attention(rt, q.float(), k.float(), k.float(), 1.0)


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-5), (torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("pos,ms", [(0, 1), (5, 16), (16, 17), (100, 128), (511, 512)])
def test_attention_kvcache_vs_oracle(rt, dt, tol, pos, ms):
    rng = np.random.default_rng(pos + 7)
    b, h, d = 2, 3, 128
    mk = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(dt).cuda()  # noqa: E731
    kc, vc, q, k, v = mk(b, h, ms, d), mk(b, h, ms, d), mk(b, h, 1, d), mk(b, h, 1, d), mk(b, h, 1, d)
    kc0, vc0 = host(kc), host(vc)
    p = torch.tensor([[pos]], dtype=torch.int32).cuda()
    y = ops.attention_kvcache(rt, kc, vc, q, k, v, p)
    want, kc_w, vc_w = R.attention_kvcache(kc0, vc0, host(q), host(k), host(v), pos)
    assert np.allclose(host(y), want, rtol=tol, atol=tol)
    assert np.array_equal(host(kc), kc_w) and np.array_equal(host(vc), vc_w)  # appended in place, nothing else touched


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-5), (torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("split", [0, 1, 2, 5, 16])
@pytest.mark.parametrize("pos,ms,d", [(0, 64, 128), (62, 64, 128), (63, 64, 128), (64, 80, 128), (700, 1024, 128), (1023, 1024, 256), (130, 4096, 128)])
def test_attention_kvcache_split_over_workgroups(rt, dt, tol, split, pos, ms, d, monkeypatch):
    """Round 5: the cache cut into G chunks over workgroups + a merge kernel (reference: attention_kvcache.cu:18-25, 118-166 splits over
    gridDim.y and merges). IROCM_KVCACHE_SPLIT forces G (0 = the one-workgroup element-wise kernel): chunk ends that fall inside an
    iteration's four keys, EMPTY chunks (n far below the capacity the split was sized for), the new key in the last / first chunk,
    head dim 256 — same oracle, same in-place append."""
    monkeypatch.setenv("IROCM_KVCACHE_SPLIT", str(split))
    rng = np.random.default_rng(pos + 7 * split)
    b, h = 2, 3
    mk = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(dt).cuda()  # noqa: E731
    kc, vc, q, k, v = mk(b, h, ms, d), mk(b, h, ms, d), mk(b, h, 1, d), mk(b, h, 1, d), mk(b, h, 1, d)
    kc0, vc0 = host(kc), host(vc)
    p = torch.tensor([[pos]], dtype=torch.int64).cuda()
    y = ops.attention_kvcache(rt, kc, vc, q, k, v, p)
    want, kc_w, vc_w = R.attention_kvcache(kc0, vc0, host(q), host(k), host(v), pos)
    assert np.allclose(host(y), want, rtol=tol, atol=tol), np.abs(host(y) - want).max()
    assert np.array_equal(host(kc), kc_w) and np.array_equal(host(vc), vc_w)  # appended in place, nothing else touched


def test_attention_kvcache_consecutive_decode_steps(rt, monkeypatch):
    """Twelve consecutive decode steps on one cache with the split forced to 7 chunks (partial results and merge of every step reuse
    the same workspace) against the oracle step by step."""
    monkeypatch.setenv("IROCM_KVCACHE_SPLIT", "7")
    rng = np.random.default_rng(11)
    b, h, ms, d = 2, 5, 1024, 128
    mk = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(torch.float16).cuda()  # noqa: E731
    kc, vc = mk(b, h, ms, d), mk(b, h, ms, d)
    kc_h, vc_h = host(kc), host(vc)
    for pos in range(600, 612):
        q, k, v = mk(b, h, 1, d), mk(b, h, 1, d), mk(b, h, 1, d)
        y = ops.attention_kvcache(rt, kc, vc, q, k, v, torch.tensor([pos], dtype=torch.int32).cuda())
        want, kc_h, vc_h = R.attention_kvcache(kc_h, vc_h, host(q), host(k), host(v), pos)
        assert np.allclose(host(y), want, rtol=2e-3, atol=2e-3), (pos, np.abs(host(y) - want).max())
    assert np.array_equal(host(kc), kc_h) and np.array_equal(host(vc), vc_h)


def test_attention_kvcache_llama_decode_shape(rt):
    """B x H = 32, 4096 cached keys, D = 128, f16 (a batch-1 Llama-7B decode step): the heuristic split (no env) against the oracle."""
    rng = np.random.default_rng(3)
    b, h, ms, d, pos = 1, 32, 4096, 128, 4095
    mk = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(torch.float16).cuda()  # noqa: E731
    kc, vc, q, k, v = mk(b, h, ms, d), mk(b, h, ms, d), mk(b, h, 1, d), mk(b, h, 1, d), mk(b, h, 1, d)
    kc0, vc0 = host(kc), host(vc)
    y = ops.attention_kvcache(rt, kc, vc, q, k, v, torch.tensor([pos], dtype=torch.int32).cuda())
    want, kc_w, vc_w = R.attention_kvcache(kc0, vc0, host(q), host(k), host(v), pos)
    assert np.allclose(host(y), want, rtol=2e-3, atol=2e-3), np.abs(host(y) - want).max()
    assert np.array_equal(host(kc), kc_w) and np.array_equal(host(vc), vc_w)


def test_attention_kvcache_reference_kat(rt):
    """test_cuda_attention.cc:17-43: ones everywhere, position 0 -> ones."""
    from conftest import kat

    one = lambda *s: torch.ones(s, dtype=torch.float32).cuda()  # noqa: E731
    y = ops.attention_kvcache(rt, torch.zeros(1, 1, 1, 128).cuda(), torch.zeros(1, 1, 1, 128).cuda(), one(1, 1, 1, 128),
                              one(1, 1, 1, 128), one(1, 1, 1, 128), torch.zeros(1, 1, dtype=torch.int32).cuda())
    assert R.equal_data(host(y).ravel(), kat("test/kernels/cuda/test_cuda_attention.cc", 36, "float"), 1e-6)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("case", CASES)
def test_attention_head_merge_store_is_the_transposed_result(rt, case, dt):
    """infini_rocm_attention_ex(heads = H): [B, Sq, H, D] output == Transpose(0, 2, 1, 3) of the plain [B, H, Sq, D] result, bit
    for bit (same kernel, only the store address differs), over the ragged / masked / causal cases above."""
    b, h, sq, sk, d, use_mask, causal = case
    rng = np.random.default_rng(abs(hash(case)) % 2 ** 32)
    qd, kd, vd = (dev(rng.standard_normal(s), TD[dt]) for s in ((b, h, sq, d), (b, h, sk, d), (b, h, sk, d)))
    md = None
    if use_mask:
        m = np.where(rng.random((b, sk)) < 0.8, 0.0, -10000.0).astype(np.float32)
        m[:, 0] = 0.0
        md = dev(m, TD[dt])
    plain = ops.attention(rt, qd, kd, vd, 1.0 / np.sqrt(d), md, causal)
    merged = ops.attention(rt, qd, kd, vd, 1.0 / np.sqrt(d), md, causal, head_merge=h)
    assert tuple(merged.shape) == (b, sq, h, d)
    assert torch.equal(merged, plain.permute(0, 2, 1, 3).contiguous())


MASK2D_CASES = [
    # b, h, sq, sk, d, mask batch dim, mask head dim
    (2, 3, 128, 128, 64, 1, 1),     # one [Sq, Sk] mask for everything (a causal mask as a tensor)
    (2, 2, 100, 77, 64, 2, 1),      # per batch entry, ragged tiles, Sk % 4 != 0 (element loads)
    (2, 2, 64, 192, 128, 2, 2),     # per (batch, head)
    (1, 4, 256, 256, 128, 1, 1),    # Llama head
    (1, 3, 33, 36, 64, 1, 3),       # batch 1 with a per-head mask
]


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("case", MASK2D_CASES)
def test_attention_full_additive_mask(rt, case, dt):
    """mask [G, Sq, Sk] (one row per query; infini_rocm_attention_ex mask_2d): random additive biases plus a causal
    pattern of large negatives, vs the oracle; and the causal-mask tensor reproduces the kernel's own causal option."""
    b, h, sq, sk, d, mb, mh = case
    rng = np.random.default_rng(abs(hash(case)) % 2 ** 32)
    q, k, v = (rng.standard_normal(s).astype(np.float32) for s in ((b, h, sq, d), (b, h, sk, d), (b, h, sk, d)))
    m = (0.5 * rng.standard_normal((mb, mh, sq, sk))).astype(np.float32)
    m = np.where(np.tril(np.ones((sq, sk), dtype=bool), k=sk - sq), m, -10000.0).astype(np.float32)
    m[..., 0] = np.minimum(m[..., 0], 0) * 0  # key 0 is always admissible: no fully masked row
    scale = 1.0 / np.sqrt(d)
    qd, kd, vd = dev(q, TD[dt]), dev(k, TD[dt]), dev(v, TD[dt])
    y = ops.attention(rt, qd, kd, vd, scale, dev(m.reshape(mb * mh, sq, sk), TD[dt]))
    want = R.attention(R.round_to(q, dt), R.round_to(k, dt), R.round_to(v, dt), scale, R.round_to(m, dt))
    assert np.allclose(host(y), want, rtol=TOL[dt], atol=TOL[dt])
    if sq == sk:
        causal_tensor = np.triu(np.full((sq, sk), -10000.0, np.float32), 1)[None]
        y_mask = ops.attention(rt, qd, kd, vd, scale, dev(causal_tensor, TD[dt]))
        y_flag = ops.attention(rt, qd, kd, vd, scale, None, True)
        assert np.allclose(host(y_mask), host(y_flag), rtol=TOL[dt], atol=TOL[dt])
