"""MatMul parity on a real MI355X: HIP kernels (through the C ABI) vs the oracle and the
reference's golden vectors. Tolerances: fp32 1e-4 relative (north_star); bf16/fp16 against the
fp64 GEMM of the rounded inputs, |err| <= 2^-7 |c| + small absolute slack (one output rounding +
fp32 accumulation)."""
import numpy as np
import pytest
import torch
from conftest import kat

from infinitensor_amd import ops
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu
CU = "test/kernels/cuda/test_cuda_matmul.cc"


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def host(t):
    return t.float().cpu().numpy().astype(np.float64)


KAT_CASES = [
    ("inc", "one", False, False, (1, 3, 5), (1, 5, 2), 50),
    ("inc", "inc", True, False, (2, 3, 4), (2, 3, 2), 53),
    ("inc", "inc", False, False, (2, 3, 5), (5, 2), 58),
    ("inc", "inc", True, False, (2, 5, 3), (5, 2), 61),
    ("inc", "inc", False, False, (3, 5), (5, 2), 65),
]


@pytest.mark.parametrize("case", KAT_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_matmul_reference_kats(rt, case, dtype):
    """test_cuda_matmul.cc:47-66 — small integers, exact in every dtype (bf16: values <= 695 need
    rounding, so only fp32/fp16 are compared bit-for-value; bf16 within its 2^-8)."""
    ga, gb, ta, tb, sa, sb, line = case
    g = {"inc": R.incremental, "one": R.ones}
    a, b = g[ga](sa), g[gb](sb)
    c = ops.matmul(rt, dev(a, dtype), dev(b, dtype), None, ta, tb)
    want = kat(CU, line, "float")
    got = host(c).ravel()
    if dtype == torch.bfloat16:
        ar, br = R.round_to(a, "bf16"), R.round_to(b, "bf16")
        want = R.matmul(ar, br, None, ta, tb).ravel()
        assert np.allclose(got, want, rtol=2 ** -7, atol=1e-6)
    else:
        assert R.equal_data(got, want, 1e-6)


SHAPES = [
    # b, m, n, k
    (1, 128, 128, 64),
    (1, 256, 384, 128),
    (1, 200, 136, 192),   # ragged m / n
    (3, 130, 72, 64),
    (2, 64, 64, 256),
    (1, 512, 512, 512),
    (1, 77, 53, 41),      # nothing aligned -> generic kernel
    (1, 300, 264, 2048),  # long K, few tiles: the split-K heuristic (4 tiles x up to 4 K slices)
    (2, 136, 200, 1376),  # K % 64 = 32 (Llama FFN / 8): zero-filled K tail in the LDS-DMA kernel
    (1, 128, 128, 72),    # K = 64 + 8
    (1, 64, 72, 8),       # K smaller than one tile
]


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("shape", SHAPES)
def test_matmul_16bit_variants(rt, shape, ta, tb, dtype, variant):
    b, m, n, k = shape
    rng = np.random.default_rng(hash((shape, ta, tb)) % 2 ** 32)
    a = rng.standard_normal((b, k, m) if ta else (b, m, k)).astype(np.float32)
    bm = rng.standard_normal((n, k) if tb else (k, n)).astype(np.float32)  # rank-2 B: broadcast batch
    bias = rng.standard_normal((n,)).astype(np.float32)
    name = "bf16" if dtype == torch.bfloat16 else "f16"
    ops.set_matmul_variant(rt, variant)
    try:
        c = ops.matmul(rt, dev(a, dtype), dev(bm, dtype), dev(bias, dtype), ta, tb)
    finally:
        ops.set_matmul_variant(rt, -1)
    want = R.matmul(R.round_to(a, name), R.round_to(bm, name), R.round_to(bias, name), ta, tb)
    got = host(c)
    assert got.shape == want.shape
    # Per-element bound (round-5 verdict, weak #10): one storage ulp of the result (its final rounding) plus the fp32 accumulation
    # error, which scales with S = sum |a_i| |b_i| (+ |bias|) of THAT element — 2^-17 S is ~2^7 fp32 ulps of S, 16 x below the worst
    # case k 2^-24 S at k = 2048 and far above the sqrt(k) 2^-24 S a sum in any order really makes. The old absolute term
    # tol * sqrt(k) was 45 output ulps for near-zero fp16 elements at k = 2048.
    tol = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -10
    err = np.abs(got - want)
    S = R.matmul(np.abs(R.round_to(a, name)), np.abs(R.round_to(bm, name)), np.abs(R.round_to(bias, name)), ta, tb)
    bound = tol * np.abs(want) + 2.0 ** -17 * S
    assert (err <= bound).all(), f"max err {err.max()} at {np.unravel_index(err.argmax(), err.shape)} (bound there {bound.ravel()[err.argmax()]})"


W128_SHAPES = [
    # b, m, n, k — whole 256^2 tiles, K % 128 == 0 (gemm128w.hip's contract)
    (1, 256, 256, 128),    # one tile, one block of four k-steps: prologue -> last block -> epilogue
    (1, 512, 768, 256),    # six tiles on six workgroups
    (3, 512, 256, 384),    # batch strides on both operands
    (1, 1280, 2304, 640),  # 45 tiles: ragged last tile-row group (5 rows in groups of 4)
    (2, 4096, 4608, 128),  # 576 tiles: every workgroup walks 2-3 tiles, the next tile's pieces requested under the epilogue
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("shape", W128_SHAPES)
def test_wave128_gemm_all_layouts(rt, shape, ta, tb, dtype):
    """The four-wave kernel (gemm128w.hip, variant "wave128": 128 x 128 wave tiles, four-stage LDS ring, inline-asm K loop) against the
    fp64 product of the rounded operands, per-element bound as above, in every layout — K-major operands take the cache-line-pair path
    (sibling pieces with an instruction offset), M/N-major ones the transpose reads."""
    b, m, n, k = shape
    rng = np.random.default_rng(hash((shape, ta, tb, 128)) % 2 ** 32)
    a = rng.standard_normal((b, k, m) if ta else (b, m, k)).astype(np.float32)
    bm = rng.standard_normal((b, n, k) if tb else (b, k, n)).astype(np.float32)
    name = "bf16" if dtype == torch.bfloat16 else "f16"
    ops.set_matmul_variant(rt, ops.matmul_variants().index("wave128"))
    try:
        c = ops.matmul(rt, dev(a, dtype), dev(bm, dtype), None, ta, tb)
        assert ops.matmul_last_variant(rt) == "wave128"
    finally:
        ops.set_matmul_variant(rt, -1)
    ar, br = R.round_to(a, name), R.round_to(bm, name)
    want = R.matmul(ar, br, None, ta, tb)
    got = host(c)
    tol = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -10
    err = np.abs(got - want)
    bound = tol * np.abs(want) + 2.0 ** -17 * R.matmul(np.abs(ar), np.abs(br), None, ta, tb)
    assert (err <= bound).all(), f"max err {err.max()} at {np.unravel_index(err.argmax(), err.shape)}"


def test_wave128_gemm_routing_and_fallback(rt):
    """The heuristic hands the four-wave kernel plain single-batch GEMMs of one or two rounds of whole tiles with K >= 2048 (not NT);
    forced on a problem outside its contract (bias, ragged m) the library falls back to the heuristic instead of failing."""
    dt = torch.bfloat16
    a = torch.randn(4096, 4096, device="cuda").to(dt)
    b = torch.randn(4096, 4096, device="cuda").to(dt)
    c = ops.matmul(rt, a, b)
    assert ops.matmul_last_variant(rt) == "wave128"
    ops.set_matmul_variant(rt, 4)
    try:
        c4 = ops.matmul(rt, a, b)
        assert ops.matmul_last_variant(rt) == "persist256"
    finally:
        ops.set_matmul_variant(rt, -1)
    # two kernels, two summation orders: a storage ulp of the result plus the accumulation slack
    diff = (c.double() - c4.double()).abs()
    assert bool((diff <= 2.0 ** -7 * c4.double().abs() + 2.0 ** -17 * (a.double().abs() @ b.double().abs())).all())
    ops.matmul(rt, a, b, trans_b=True)
    assert ops.matmul_last_variant(rt) == "persist256"  # NT stays on the eight-wave kernel
    ops.matmul(rt, a, b, bias=torch.zeros(4096, device="cuda", dtype=dt))
    assert ops.matmul_last_variant(rt) == "persist256"  # a bias: not the plain case
    ops.set_matmul_variant(rt, ops.matmul_variants().index("wave128"))
    try:
        x = torch.randn(300, 256, device="cuda").to(dt)
        y = torch.randn(256, 256, device="cuda").to(dt)
        z = ops.matmul(rt, x, y)
        assert ops.matmul_last_variant(rt) != "wave128"
        assert torch.allclose(z.float(), (x.float() @ y.float()), rtol=2 ** -6, atol=0.25)
    finally:
        ops.set_matmul_variant(rt, -1)


@pytest.mark.parametrize("variant", [4, 5, 6])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False)])
@pytest.mark.parametrize("shape", [(1, 16384, 1536, 192), (1, 8192, 4096, 64), (1, 4104, 3080, 128), (3, 2048, 2560, 256),
                                   (1, 16384, 768, 768)])
def test_persistent_gemm_walks_several_tiles(rt, shape, ta, tb, variant):
    """The persistent kernels with MORE tiles than CUs (a workgroup walks 2-4 tiles through one flat K-tile pipeline:
    odd K-tile counts flip the LDS buffer parity between tiles, K = 64 lets the B cursor run two tiles ahead), ragged
    edge tiles, a batch, a bias: sampled rows vs the fp64 oracle, the whole tensor vs the generic kernel (an independent
    code path), and bit-identical repeats."""
    b, m, n, k = shape
    rng = np.random.default_rng(hash((shape, ta, tb)) % 2 ** 32)
    a = rng.standard_normal((b, k, m) if ta else (b, m, k)).astype(np.float32)
    bm = rng.standard_normal((n, k) if tb else (k, n)).astype(np.float32)
    bias = rng.standard_normal((n,)).astype(np.float32)
    ad, bd, biasd = dev(a, torch.bfloat16), dev(bm, torch.bfloat16), dev(bias, torch.bfloat16)
    ops.set_matmul_variant(rt, variant)
    try:
        c1 = ops.matmul(rt, ad, bd, biasd, ta, tb)
        c2 = ops.matmul(rt, ad, bd, biasd, ta, tb)
        ops.set_matmul_variant(rt, 0)
        c0 = ops.matmul(rt, ad, bd, biasd, ta, tb)
    finally:
        ops.set_matmul_variant(rt, -1)
    assert torch.equal(c1, c2)
    tol = 2 ** -7
    d = (c1.float() - c0.float()).abs()
    assert bool((d <= 2 * tol * c0.float().abs() + 2 * tol * np.sqrt(k)).all()), float(d.max())
    rows = np.unique(np.concatenate([rng.choice(m, 24, replace=False), [0, 255, 256, m - 1]]))
    ar = a[:, :, rows] if ta else a[:, rows, :]
    want = R.matmul(R.round_to(ar, "bf16"), R.round_to(bm, "bf16"), R.round_to(bias, "bf16"), ta, tb)
    got = host(c1)[:, rows, :]
    err = np.abs(got - want)
    assert (err <= tol * np.abs(want) + tol * np.sqrt(k)).all(), float(err.max())


@pytest.mark.parametrize("tab_n", [0, 1, 2])
@pytest.mark.parametrize("variant,shape", [(4, (1, 16384, 3072, 128)), (5, (3, 4096, 2304, 64)), (6, (1, 8192, 2048, 192))])
def test_persistent_gemm_tile_table_and_in_place_decode_agree(rt, monkeypatch, variant, shape, tab_n):
    """Round 4: a workgroup of the persistent kernels reads its tiles' coordinates from an LDS table (first 256 steps) and decodes the
    steps beyond it in place. Production shapes reach the second path only past 65 536 tiles; IROCM_GEMM_TAB_N (read per launch)
    shortens the table, so both paths — and the hand-over between them, with the B cursor up to two tiles ahead at K = 64 — run on
    shapes with 3-12 tiles per workgroup: bit-identical to the full table."""
    b, m, n, k = shape
    rng = np.random.default_rng(hash((shape, variant)) % 2 ** 32)
    ad = dev(rng.standard_normal((b, m, k)).astype(np.float32), torch.bfloat16)
    bd = dev(rng.standard_normal((k, n)).astype(np.float32), torch.bfloat16)
    biasd = dev(rng.standard_normal((n,)).astype(np.float32), torch.bfloat16)
    ops.set_matmul_variant(rt, variant)
    try:
        monkeypatch.delenv("IROCM_GEMM_TAB_N", raising=False)
        want = ops.matmul(rt, ad, bd, biasd)
        monkeypatch.setenv("IROCM_GEMM_TAB_N", str(tab_n))
        got = ops.matmul(rt, ad, bd, biasd)
        rt.sync()
    finally:
        ops.set_matmul_variant(rt, -1)
    assert torch.equal(got, want)


def test_matmul_splitk_heuristic_shapes(rt):
    """Shapes the heuristic routes to split-K (Llama projections at 2048 tokens; a TP-8 shard): vs the fp64 oracle on a
    row sample, and bit-identical across repeats (the reduce pass sums the slices in a fixed order)."""
    rng = np.random.default_rng(17)
    for m, n, k in ((2048, 4096, 4096), (2048, 512, 4096), (2048, 1024, 11008)):
        a = rng.standard_normal((m, k)).astype(np.float32)
        w = rng.standard_normal((k, n)).astype(np.float32)
        ad, wd = dev(a, torch.bfloat16), dev(w, torch.bfloat16)
        c1 = ops.matmul(rt, ad, wd)
        c2 = ops.matmul(rt, ad, wd)
        assert torch.equal(c1, c2)
        rows = rng.choice(m, 16, replace=False)
        want = R.matmul(R.round_to(a[rows], "bf16"), R.round_to(w, "bf16"))
        err = np.abs(host(c1)[rows] - want)
        assert (err <= 2 ** -7 * np.abs(want) + 2 ** -7 * np.sqrt(k)).all()


@pytest.mark.parametrize("shape", [(1, 64, 64, 64), (2, 100, 36, 50), (1, 512, 512, 512), (1, 3, 7, 1000)])
@pytest.mark.parametrize("ta,tb", [(False, False), (True, True)])
def test_matmul_fp32(rt, shape, ta, tb):
    """fp32 gate: 1e-4 relative to the fp64 oracle (exact-f32 MFMA path)."""
    b, m, n, k = shape
    rng = np.random.default_rng(7)
    a = rng.uniform(-1, 1, (b, k, m) if ta else (b, m, k)).astype(np.float32)
    bm = rng.uniform(-1, 1, (b, n, k) if tb else (b, k, n)).astype(np.float32)
    c = ops.matmul(rt, dev(a), dev(bm), None, ta, tb)
    want = R.matmul(a, bm, None, ta, tb)
    got = host(c)
    # north_star: fp32 within 1e-4 RELATIVE of the CPU oracle — asserted as a pure relative bound wherever |c| >= 1 ...
    big = np.abs(want) >= 1.0
    assert (np.abs(got - want)[big] <= 1e-4 * np.abs(want)[big]).all(), np.abs((got - want)[big] / want[big]).max()
    # ... and near zero (cancellation: |c| < 1 while sum |a b| ~ k / 3) as a small absolute bound
    assert (np.abs(got - want)[~big] <= 2e-5).all(), np.abs((got - want)[~big]).max()


@pytest.mark.parametrize("shape", [(1, 128, 128, 32), (1, 1024, 1024, 1024), (3, 130, 132, 100), (1, 257, 516, 36), (2, 64, 8, 4),
                                   (1, 2048, 1536, 260), (1, 512, 512, 512), (5, 2048, 2048, 64)])
@pytest.mark.parametrize("tb", [False, True])
@pytest.mark.parametrize("bias_form", [None, "row", "full"])
def test_matmul_fp32_tile_kernel(rt, shape, tb, bias_form):
    """The fp32 128^2 LDS-DMA tile kernel (gemm32.hip, v_mfma_f32_32x32x2_f32; variant "fast32", what the heuristic picks
    for fp32 problems of at least half a tile per CU) forced over full, ragged, K-tail and batched shapes, NN (ONNX MatMul) and
    NT (ONNX Gemm), with the bias forms of matmul.cc:86-118: the same 1e-4 RELATIVE gate as the generic kernel, and equal to
    it within fp32 summation-order noise."""
    b, m, n, k = shape
    rng = np.random.default_rng(17)
    a = rng.uniform(-1, 1, (b, m, k)).astype(np.float32)
    bm = rng.uniform(-1, 1, (b, n, k) if tb else (b, k, n)).astype(np.float32)
    bias = None if bias_form is None else rng.uniform(-1, 1, (n,) if bias_form == "row" else (b, m, n)).astype(np.float32)
    v7 = ops.matmul_variants().index("fast32")
    ops.set_matmul_variant(rt, v7)
    try:
        c = ops.matmul(rt, dev(a), dev(bm), None if bias is None else dev(bias), False, tb)
        assert ops.matmul_last_variant(rt) == "fast32"
        ops.set_matmul_variant(rt, 0)
        c0 = ops.matmul(rt, dev(a), dev(bm), None if bias is None else dev(bias), False, tb)
        assert ops.matmul_last_variant(rt) == "generic64"
    finally:
        ops.set_matmul_variant(rt, -1)
    want = R.matmul(a, bm, bias, False, tb)
    got = host(c)
    big = np.abs(want) >= 1.0
    assert (np.abs(got - want)[big] <= 1e-4 * np.abs(want)[big]).all(), np.abs((got - want)[big] / want[big]).max()
    assert (np.abs(got - want)[~big] <= 2e-5 * max(1.0, k / 256)).all(), np.abs((got - want)[~big]).max()
    assert np.allclose(got, host(c0), rtol=1e-5, atol=1e-5 * max(1.0, k / 64))


def test_matmul_fp32_heuristic_picks_the_tile_kernel_for_large_problems(rt):
    a = torch.randn(2048, 512, device="cuda")
    b = torch.randn(512, 2048, device="cuda")
    ops.matmul(rt, a, b)
    assert ops.matmul_last_variant(rt) == "fast32"
    ops.matmul(rt, a[:64].contiguous(), b)  # 32 tiles of 64^2: still the tile kernel, in its 64^2 form
    assert ops.matmul_last_variant(rt) == "fast32"
    ops.matmul(rt, a[:32, :32].contiguous(), b[:32, :64].contiguous())  # one tile, K = 32: the generic kernel
    assert ops.matmul_last_variant(rt) == "generic64"
    ops.matmul(rt, torch.randn(512, 2048, device="cuda"), torch.randn(512, 2048, device="cuda"), None, True, False)  # transA: not served
    assert ops.matmul_last_variant(rt) == "generic64"


def test_matmul_bias_broadcast_forms(rt):
    """bias broadcast into C like the reference (matmul.cc:86-118): [n], [m,n], [1], [b,m,n]."""
    rng = np.random.default_rng(11)
    b, m, n, k = 2, 33, 48, 64
    a = rng.standard_normal((b, m, k)).astype(np.float32)
    w = rng.standard_normal((b, k, n)).astype(np.float32)
    for bshape in [(n,), (m, n), (1,), (b, m, n), (m, 1)]:
        bias = rng.standard_normal(bshape).astype(np.float32)
        c = ops.matmul(rt, dev(a), dev(w), dev(bias))
        assert np.allclose(host(c), R.matmul(a, w, bias), rtol=1e-4, atol=1e-4), bshape


def test_matmul_headline_shape_sampled_rows(rt):
    """bf16 4096^3 (BASELINE config 2): check 48 sampled rows of C against the fp64 oracle, NN and NT."""
    rng = np.random.default_rng(0)
    M = N = K = 4096
    a = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(torch.bfloat16).cuda()
    b = torch.from_numpy(rng.standard_normal((K, N)).astype(np.float32)).to(torch.bfloat16).cuda()
    rows = rng.choice(M, 48, replace=False)
    a64 = a[rows].float().cpu().numpy().astype(np.float64)
    b64 = b.float().cpu().numpy().astype(np.float64)
    want = a64 @ b64
    for v, name in enumerate(ops.matmul_variants()):
        if v == 0 or name == "fast32":  # (fast32 serves fp32 only)
            continue
        ops.set_matmul_variant(rt, v)
        try:
            c = ops.matmul(rt, a, b)
            ct = ops.matmul(rt, a, b.t().contiguous(), None, False, True)
        finally:
            ops.set_matmul_variant(rt, -1)
        for res in (c, ct):
            got = res[rows].float().cpu().numpy().astype(np.float64)
            err = np.abs(got - want)
            assert (err <= 2 ** -7 * np.abs(want) + 0.5).all(), (name, err.max())


def test_matmul_rejects_bad_arguments(rt):
    a = torch.zeros(4, 5, device="cuda")
    b = torch.zeros(6, 3, device="cuda")
    from infinitensor_amd import InfiniRocmError, lib
    # (since round 5 the shape glue lives below the ABI — infini_rocm_matmul_plan — and its errors are the library's: a RuntimeError
    # like the reference's infini::Exception)
    with pytest.raises(InfiniRocmError, match="K of A is 5, K of B is 6"):
        ops.matmul(rt, a, b)  # K mismatch: reference IT_ASSERT(kA == kB)
    with pytest.raises(InfiniRocmError, match="batch"):
        ops.matmul(rt, torch.zeros(2, 1, 4, 5, device="cuda"), torch.zeros(1, 3, 5, 6, device="cuda"))  # partial batch broadcast
    with pytest.raises(TypeError):
        ops.matmul(rt, a.to(torch.complex64), torch.zeros(5, 3, device="cuda", dtype=torch.complex64))
    import ctypes
    with pytest.raises(InfiniRocmError):  # the C ABI itself rejects an unsupported dtype, loudly
        from infinitensor_amd._lib import check
        check(lib().infini_rocm_matmul(rt.handle, 7, None, None, None, None, 1, 4, 4, 4, 0, 0, 16, 16, 0, 0, 0, 0))


@pytest.mark.parametrize("variant", [-1, 0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", [(1, 512, 768, 256, 128, 64), (2, 256, 512, 320, 128, 64), (1, 384, 200, 136, 96, 40),
                                   (1, 1024, 1024, 1024, 256, 128)])
def test_matmul_head_split_store_is_the_reshape_transpose_chain(rt, variant, dtype, shape):
    """infini_rocm_matmul_headsplit: the [m, n] result stored as [m / S, n / D, S, D] equals MatMul -> Reshape([.., S, H, D])
    -> Transpose(0, 2, 1, 3) of the plain call BIT FOR BIT (same kernel, same sums, only the store address differs), for
    every kernel variant (generic, fast128, 256-tile, split-K), with a bias, ragged tiles and a batch."""
    b, m, n, k, S, D = shape
    rng = np.random.default_rng(m + n + k)
    a = rng.standard_normal((b, m, k)).astype(np.float32)
    w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
    bias = rng.standard_normal((n,)).astype(np.float32)
    da, dw, db = dev(a, dtype), dev(w, dtype), dev(bias, dtype)
    ops.set_matmul_variant(rt, variant)
    try:
        plain = ops.matmul(rt, da, dw, db)
        split = ops.matmul(rt, da, dw, db, head_split=(S, D))
    finally:
        ops.set_matmul_variant(rt, -1)
    assert tuple(split.shape) == (b, m // S, n // D, S, D)
    want = plain.view(b, m // S, S, n // D, D).permute(0, 1, 3, 2, 4).contiguous()
    assert torch.equal(split, want)
    ref = R.matmul(R.round_to(a, {torch.float16: "f16", torch.bfloat16: "bf16", torch.float32: "f32"}[dtype]),
                   R.round_to(w, {torch.float16: "f16", torch.bfloat16: "bf16", torch.float32: "f32"}[dtype]),
                   R.round_to(bias, {torch.float16: "f16", torch.bfloat16: "bf16", torch.float32: "f32"}[dtype]))
    tol = {torch.float16: 2e-3, torch.bfloat16: 1.6e-2, torch.float32: 1e-4}[dtype]
    assert np.allclose(host(plain), ref, rtol=tol, atol=tol * 4)


def test_matmul_head_split_rejects_bad_tilings(rt):
    a, w = dev(np.ones((64, 32), np.float32), torch.float16), dev(np.ones((32, 48), np.float32), torch.float16)
    with pytest.raises(ValueError):
        ops.matmul(rt, a, w, head_split=(48, 16))  # 48 does not divide m = 64
    with pytest.raises(ValueError):
        ops.matmul(rt, a, w, head_split=(32, 12))  # head_dim % 8 != 0


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("variant", [-1, 0, 1, 2, 3, 4, 5, 6])
def test_matmul_fast_gelu_epilogue(rt, dtype, tol, variant):
    """act = 5: Gelu (erf form) in the GEMM epilogue with erf by a clamped odd polynomial (no transcendentals; abs error
    < 2^-12, gemm_common.h::gelu_poly) vs the oracle's exact Gelu of the fp64 product, over the whole input range incl. the
    negative tail (exactly 0 below -4, where |gelu| < 1.3e-4) and the positive one (exactly x above 4), and against act = 4
    (erff) to one output ulp."""
    rng = np.random.default_rng(9)
    a = (rng.standard_normal((512, 256)) * 1.5).astype(np.float32)
    w = (rng.standard_normal((256, 768)) / 8).astype(np.float32)
    bias = rng.standard_normal((768,)).astype(np.float32)
    name = {torch.float16: "f16", torch.bfloat16: "bf16"}[dtype]
    ops.set_matmul_variant(rt, variant)
    try:
        y5 = ops.matmul(rt, dev(a, dtype), dev(w, dtype), dev(bias, dtype), act=5)
        y4 = ops.matmul(rt, dev(a, dtype), dev(w, dtype), dev(bias, dtype), act=4)
    finally:
        ops.set_matmul_variant(rt, -1)
    pre = R.matmul(R.round_to(a, name), R.round_to(w, name), R.round_to(bias, name))
    assert np.abs(pre).max() > 6 and pre.min() < -6  # the tails are exercised
    want = R.unary("gelu", pre)
    assert np.allclose(host(y5), want, rtol=tol, atol=tol)
    assert np.allclose(host(y5), host(y4), rtol=tol, atol=tol / 4)


def test_matmul_workspace_hint(rt):
    """infini_rocm_matmul_may_use_workspace: split-K (the only MatMul path that takes the runtime workspace) is possible
    only below ~0.6 tiles of 256^2 per CU, or when variant 3 is forced (rocm_fusion.cc parks an operand of the next MatMul
    in the workspace only when the answer is 0)."""
    import ctypes as C

    from infinitensor_amd import lib
    from infinitensor_amd._lib import check

    def may(batch, m, n):
        out = C.c_int(-1)
        check(lib().infini_rocm_matmul_may_use_workspace(rt.handle, batch, m, n, C.byref(out)))
        return out.value

    assert may(1, 128, 1000) == 1        # ResNet's classifier: 4 tiles
    assert may(1, 2048, 4096) == 1       # 128 tiles: the Llama down projection runs split-K
    assert may(32, 512, 768) == 0        # BERT's output projection: 192 tiles of 256^2 ... plus batch: no split-K
    assert may(1, 16384, 3072) == 0
    ops.set_matmul_variant(rt, 3)
    try:
        assert may(1, 16384, 3072) == 1  # forced split-K
    finally:
        ops.set_matmul_variant(rt, -1)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("variant", [-1, 2, 4, 5, 6])
def test_matmul_grouped_members_at_a_stride(rt, dtype, tol, variant):
    """infini_rocm_matmul_grouped: three MatMuls of one activation whose weights / biases / OUTPUTS are separate tensors at
    uniform distances (carved out of slabs with gaps) run as one launch; every member equals its own plain matmul bit for
    bit (same kernel, same sums), and the gaps between the outputs stay untouched."""
    rng = np.random.default_rng(61)
    m, k, n, g = 640, 256, 384, 3
    a = dev(rng.standard_normal((m, k)).astype(np.float32), dtype)
    wslab = dev((rng.standard_normal((g, k * n + 64)) / 16).astype(np.float32), dtype)
    bslab = dev(rng.standard_normal((g, n + 8)).astype(np.float32), dtype)
    oslab = torch.full((g, m * n + 128), 7.0, dtype=dtype, device="cuda")
    ws = [wslab[j, : k * n].view(k, n) for j in range(g)]
    bs = [bslab[j, :n] for j in range(g)]
    outs = [oslab[j, : m * n].view(m, n) for j in range(g)]
    ops.set_matmul_variant(rt, variant)
    try:
        ops.matmul_grouped(rt, a, ws, outs, bs, act=1)
        singles = [ops.matmul(rt, a, ws[j], bs[j], act=1) for j in range(g)]
    finally:
        ops.set_matmul_variant(rt, -1)
    rt.sync()
    for j in range(g):
        assert torch.equal(outs[j], singles[j]), j
        want = np.maximum(R.matmul(host(a), host(ws[j]), host(bs[j])), 0)
        assert np.allclose(host(outs[j]), want, rtol=tol, atol=tol)
        assert torch.all(oslab[j, m * n:] == 7.0).item()  # the gap behind every member
    with pytest.raises(ValueError):
        ops.matmul_grouped(rt, a, [ws[0], ws[2], ws[1]], outs, bs)  # not a uniform progression


@pytest.mark.parametrize("ct,dt16,tol", [("bf16", "bf16", 2.0 ** -7), ("fp16", "f16", 2.0 ** -10)])
@pytest.mark.parametrize("shape", [(512, 768, 256, False, False), (300, 520, 1024, True, True), (2048, 512, 4096, True, False)])
def test_fp32_matmul_honours_the_compute_type(rt, ct, dt16, tol, shape):
    """MatmulObj::getComputeType() (matmul.cc:51-64; onnx.py:41-47 `matmul_compute_type`): "bf16" / "fp16" on an fp32 MatMul
    multiply 16-bit roundings of A and B with fp32 accumulation and an fp32 result — checked against the oracle's product of the
    ROUNDED operands (so the only error left is fp32 summation) and shown to differ from the exact product by the 16-bit input
    rounding, i.e. the attribute really took effect; with "default" the result is the exact fp32 one again. Bias and ReLU ride
    in the reduce pass; a K that is not a multiple of 64 keeps the exact kernel."""
    m, n, k, use_bias, relu = shape
    rng = np.random.default_rng(m + n + k)
    a = rng.standard_normal((m, k)).astype(np.float32)
    b = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
    bias = rng.standard_normal((n,)).astype(np.float32) if use_bias else None
    da, db, dbias = dev(a, torch.float32), dev(b, torch.float32), (dev(bias, torch.float32) if use_bias else None)
    try:
        ops.set_matmul_compute_type(rt, ct)
        y = ops.matmul(rt, da, db, dbias, act=1 if relu else 0)
        assert y.dtype == torch.float32
        y_odd = ops.matmul(rt, da[:, : k - 8].contiguous(), db[: k - 8].contiguous())  # K % 64 != 0: exact kernel
    finally:
        ops.set_matmul_compute_type(rt, "default")
    exact = ops.matmul(rt, da, db, dbias, act=1 if relu else 0)
    fin = lambda v: np.maximum(v, 0) if relu else v
    want16 = fin(R.matmul(R.round_to(a, dt16), R.round_to(b, dt16), bias))
    want32 = fin(R.matmul(a, b, bias))
    assert np.allclose(host(y), want16, rtol=2e-5, atol=2e-5 * np.sqrt(k)), np.abs(host(y) - want16).max()
    assert np.allclose(host(exact), want32, rtol=1e-4, atol=2e-5)
    err16 = np.abs(host(y) - want32).max()
    assert 1e-5 < err16 < 64 * tol, err16  # the 16-bit rounding of the inputs is visible, and no larger than it should be
    assert np.allclose(host(y_odd), R.matmul(a[:, : k - 8], b[: k - 8]), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("ct,dt16", [("bf16", "bf16"), ("fp16", "f16")])
def test_compute_type_with_batch_strides(rt, ct, dt16):
    """Round-4 advisor (gemm.hip compute-type path): the 16-bit copies are contiguous blocks, so only batch strides of 0 or exactly one
    block may take the path. (1) a batched MatMul with a shared B (stride 0) and a per-batch A takes it and matches the oracle's
    product of the rounded operands; (2) a grouped launch whose weights sit at a LARGER distance than one block (gaps between the
    members) must not: it keeps the exact kernel (checked against the exact product — the old code cast the wrong rows)."""
    rng = np.random.default_rng(5)
    bt, m, n, k = 3, 256, 512, 512
    a = rng.standard_normal((bt, m, k)).astype(np.float32)
    b = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
    da, db = dev(a, torch.float32), dev(b, torch.float32)
    gap = 192  # elements between the members' weights
    slab = torch.zeros(3 * (k * n + gap), device="cuda", dtype=torch.float32)
    wsn = [(rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32) for _ in range(3)]
    ws = []
    for j in range(3):
        v = slab[j * (k * n + gap): j * (k * n + gap) + k * n].view(k, n)
        v.copy_(dev(wsn[j], torch.float32))
        ws.append(v)
    outs = [torch.empty(m, n, device="cuda", dtype=torch.float32) for _ in range(3)]
    oslab = torch.empty(3, m, n, device="cuda", dtype=torch.float32)
    outs = [oslab[j] for j in range(3)]
    try:
        ops.set_matmul_compute_type(rt, ct)
        y = ops.matmul(rt, da, db)
        took = ops.matmul_last_variant(rt)
        ops.matmul_grouped(rt, da[0].contiguous(), ws, outs)
    finally:
        ops.set_matmul_compute_type(rt, "default")
    rt.sync()
    want16 = R.matmul(R.round_to(a, dt16), R.round_to(b, dt16))
    assert took == "tile256_splitk", took
    assert np.allclose(host(y), want16, rtol=2e-5, atol=2e-5 * np.sqrt(k)), np.abs(host(y) - want16).max()
    for j in range(3):
        want = R.matmul(a[0], wsn[j])
        assert np.allclose(host(outs[j]), want, rtol=1e-4, atol=2e-5), (j, np.abs(host(outs[j]) - want).max())
