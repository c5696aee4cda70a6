"""Conv2d / Reduce / BatchNorm / Pool parity on a real MI355X vs the oracle and the reference KATs."""
import numpy as np
import pytest
import torch
from conftest import kat

from infinitensor_amd import ops
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu
CU = "test/kernels/cuda/"
TD = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t.to(dtype) if dtype is not None else t).cuda()


def dev_slack(a, dtype):
    """`a` on the device with 64 spare elements behind it (what every tensor of the plugin's arena has): the pixel-slot GEMM reads
    up to 14 bytes past a ragged last plane and refuses inputs for which it cannot prove those bytes readable."""
    a = np.ascontiguousarray(a)
    buf = torch.zeros((a.size + 64,), device="cuda", dtype=dtype)
    return buf[: a.size].view(a.shape).copy_(torch.from_numpy(a))


def host(t):
    return t.float().cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("g,line", [(R.ones, 50), (R.incremental, 53)])
def test_conv_reference_kats_fp32(rt, g, line):
    """test_cuda_conv.cc:48-54 (== test_mkl_conv.cc:32-37)."""
    y = ops.conv2d(rt, dev(g((1, 3, 4, 4))), dev(g((2, 3, 3, 3))), 1, 1, 2, 1, 1, 2)
    assert tuple(y.shape) == (1, 2, 2, 2)
    assert R.equal_data(host(y).ravel(), kat(CU + "test_cuda_conv.cc", line, "float"), 1e-6)


def test_conv_reference_kat_fp16(rt):
    """test_cuda_conv_fp16.cc:50-53. The reference's IncrementalGenerator fills fp16 tensors with 2.0
    (include/utils/data_generator.h:45-51), so the inputs are all-2 and the answer is 4x the ones case."""
    two = lambda s: dev(R.value(s, 2.0, np.float16))
    y = ops.conv2d(rt, two((1, 3, 4, 4)), two((2, 3, 3, 3)), 1, 1, 2, 1, 1, 2)
    assert R.equal_data(host(y).ravel(), kat(CU + "test_cuda_conv_fp16.cc", 52, "float"), 1e-6)


CONVS = [
    # n, c, h, w, f, cpg, r, s, ph, pw, sh, sw, dh, dw
    (2, 64, 56, 56, 64, 64, 1, 1, 0, 0, 1, 1, 1, 1),     # ResNet 1x1 (GEMM route)
    (2, 64, 28, 28, 128, 64, 3, 3, 1, 1, 1, 1, 1, 1),    # ResNet 3x3
    (2, 3, 64, 64, 64, 3, 7, 7, 3, 3, 2, 2, 1, 1),       # stem 7x7/2 (K = 147, ragged)
    (2, 128, 14, 14, 256, 128, 1, 1, 0, 0, 1, 1, 1, 1),  # 1x1 on 14x14 (196 pixels, not 16-B rows)
    (1, 256, 14, 14, 128, 256, 1, 1, 0, 0, 2, 2, 1, 1),  # strided 1x1 (downsample)
    (3, 32, 9, 11, 48, 8, 3, 2, 2, 0, 2, 1, 1, 2),       # groups = 4, asymmetric everything
    (1, 16, 7, 7, 24, 16, 3, 3, 1, 1, 1, 1, 1, 1),       # 7x7 plane
    (1, 512, 7, 7, 130, 512, 3, 3, 1, 1, 1, 1, 1, 1),    # ragged filter count
]


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("cfg", CONVS)
def test_conv_vs_oracle(rt, cfg, dt):
    n, c, h, w, f, cpg, r, s, ph, pw, sh, sw, dh, dw = cfg
    rng = np.random.default_rng(abs(hash(cfg)) % 2 ** 32)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = (rng.standard_normal((f, cpg, r, s)) / np.sqrt(cpg * r * s)).astype(np.float32)
    y = ops.conv2d(rt, dev(x, TD[dt]), dev(wt, TD[dt]), ph, pw, sh, sw, dh, dw)
    want = R.conv2d(R.round_to(x, dt), R.round_to(wt, dt), ph, pw, sh, sw, dh, dw)
    assert tuple(y.shape) == want.shape
    tol = {"f32": 1e-4, "f16": 2e-3, "bf16": 1.6e-2}[dt]
    assert np.allclose(host(y), want, rtol=tol, atol=tol)


S1_CONVS = [
    # unit-stride same-size shapes of conv_s1.hip: n, c, h, w, f, r, s
    (5, 64, 7, 7, 64, 3, 3),      # 49-pixel planes: slots padded to 56, tile spans images, scalar stores
    (3, 128, 14, 14, 200, 3, 3),  # 196-pixel planes, ragged filter count (two 128-row tiles)
    (2, 32, 5, 9, 40, 3, 3),      # C = 32 -> BK 32 tile, odd plane
    (2, 96, 12, 10, 72, 5, 5),    # 5x5 pad 2, C % 64 != 0
    (1, 64, 1, 1, 16, 3, 3),      # 1x1 plane: every tap but the centre is padding
    (2, 64, 3, 17, 130, 1, 7),    # 1x7 window
    (4, 256, 14, 14, 64, 1, 1),   # pointwise, F <= 64 tile
    (1, 64, 56, 56, 64, 3, 3),    # ResNet stage-1 3x3
    (130, 64, 2, 2, 8, 3, 3),     # many tiny images per tile
    # ResNet stage-4 shapes: 7x7 planes (49 pixels: rows start on 2-byte boundaries, the LDS-staged epilogue stores them
    # with unaligned 16-byte runs + a 1-pixel tail per row)
    (4, 512, 7, 7, 2048, 1, 1), (3, 2048, 7, 7, 512, 1, 1), (2, 512, 7, 7, 512, 3, 3), (3, 512, 14, 14, 512, 3, 3, 2, 2, 1, 1),
    (2, 64, 5, 3, 136, 1, 1),     # 15-pixel planes, ragged filter tile
    # conv_patch_kernel (unit-stride "same" R x S, C % 32 == 0, F > 64: input patch resident in LDS, taps = row offsets)
    (2, 128, 56, 56, 128, 3, 3),  # W = 56: halo 57 -> 64, the largest patch (256 slots)
    (9, 256, 7, 7, 136, 3, 3),    # 49-pixel planes, tiles span 2-3 images, ragged filter count, the tensor ends mid-run
    (1, 32, 60, 60, 65, 3, 3),    # one channel block, W = 60, a single valid row in the second filter tile
    (2, 64, 64, 64, 128, 3, 3),   # W = 64: the patch would need 272 slots -> the tap-shifted kernel
    (3, 160, 9, 11, 100, 3, 1),   # 3 x 1 window, 5 channel blocks, 99-pixel planes
    (2, 64, 28, 28, 512, 3, 3),   # four filter tiles share each patch position
    # pointwise, C <= 256, F > 64: conv_pw_kernel (input tile resident in LDS, loop over filter tiles)
    (3, 64, 8, 8, 256, 1, 1),      # one K-step per filter tile, two filter tiles, 1.5 column tiles
    (2, 128, 14, 14, 200, 1, 1),   # two K-steps, ragged filter count, 196-pixel planes (pad slots, tile spans images)
    (5, 256, 14, 14, 1024, 1, 1),  # ResNet 256 -> 1024 @14x14: four K-steps x eight filter tiles
    (1, 192, 6, 5, 65, 1, 1),      # C = 192 (three K-steps), a single valid row in the second filter tile, tiny plane
    (2, 256, 28, 28, 512, 1, 1, 2, 2, 1, 1),  # 1x1/2 down-sample: pointwise on the phase plane
    # strided: phase planes (conv_phase_split); n, c, h, w, f, r, s, sh, sw, dh, dw
    (3, 64, 14, 14, 96, 3, 3, 2, 2, 1, 1),    # ResNet 3x3/2 (all four phases)
    (2, 128, 28, 28, 256, 1, 1, 2, 2, 1, 1),  # ResNet 1x1/2 down-sample (one phase)
    (2, 32, 7, 9, 40, 3, 3, 2, 2, 1, 1),      # odd extents: last phase row / column is padding
    (2, 32, 9, 10, 24, 5, 3, 3, 2, 1, 1),     # stride 3 x 2
    (2, 64, 11, 11, 70, 3, 3, 1, 1, 2, 2),    # dilation 2, same size (pad 2)
    (1, 32, 13, 8, 16, 7, 7, 2, 2, 1, 1),     # 7x7/2 pad 3
    # ROWTAP (C % 32 != 0): flat K = tap * C + c padded to 32
    (2, 3, 32, 32, 64, 7, 7, 2, 2, 1, 1),     # ResNet stem
    (2, 48, 9, 9, 20, 3, 3),                  # 48 channels
    (2, 5, 11, 7, 70, 3, 5, 2, 1, 1, 1),      # 5 channels, 70 filters (two filter tiles), stride 2 x 1
    (3, 1, 6, 6, 8, 1, 1),                    # K = 1
    (1, 1, 2, 2, 4, 3, 3),                    # 8-byte input: falls back to the generic kernel
    (1, 2, 3, 5, 4, 3, 3),                    # 60-byte input
]


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("cfg", S1_CONVS)
def test_conv_s1_vs_oracle_and_generic_kernel(rt, cfg, dt):
    n, c, h, w, f, r, s, sh, sw, dh, dw = cfg + (1, 1, 1, 1) if len(cfg) == 7 else cfg
    ph, pw = (r - 1) * dh // 2, (s - 1) * dw // 2
    rng = np.random.default_rng(abs(hash(cfg)) % 2 ** 32)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = (rng.standard_normal((f, c, r, s)) / np.sqrt(c * r * s)).astype(np.float32)
    b = rng.standard_normal((f,)).astype(np.float32)
    xd, wd, bd = dev(x, TD[dt]), dev(wt, TD[dt]), dev(b, TD[dt])
    try:
        ops.set_conv_variant(rt, 2)
        y = ops.conv2d(rt, xd, wd, ph, pw, sh, sw, dh, dw, bias=bd, act=1)
        ops.set_conv_variant(rt, 1)
        yg = ops.conv2d(rt, xd, wd, ph, pw, sh, sw, dh, dw, bias=bd, act=1)
    finally:
        ops.set_conv_variant(rt, -1)
    want = np.maximum(R.conv2d(R.round_to(x, dt), R.round_to(wt, dt), ph, pw, sh, sw, dh, dw) + R.round_to(b, dt).reshape(1, f, 1, 1), 0)
    tol = {"f16": 2e-3, "bf16": 1.6e-2}[dt]
    assert np.allclose(host(y), want, rtol=tol, atol=tol)
    # same products, fp32 accumulation in a different order: the two kernels agree to rounding of the output type
    assert np.allclose(host(y), host(yg), rtol=tol, atol=tol)


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
def test_lrn_vs_oracle(rt, dt):
    """ONNX LRN across channels: hand-computed case + random shapes vs the oracle (window clipped at both channel ends,
    odd and even sizes, size > C)."""
    x = np.array([1.0, 2.0, 3.0], np.float32).reshape(1, 3, 1, 1)
    y = host(ops.lrn(rt, dev(x, TD[dt]), 3, alpha=3.0, beta=1.0, bias=1.0))
    # windows {1,2}, {1,2,3}, {2,3}: 1 / (1 + 5), 2 / (1 + 14), 3 / (1 + 13)
    tol = {"f32": 1e-6, "f16": 1e-3, "bf16": 8e-3}[dt]
    assert np.allclose(y.ravel(), [1 / 6, 2 / 15, 3 / 14], rtol=tol, atol=tol)
    rng = np.random.default_rng(4)
    for shape, size in (((2, 7, 3, 5), 5), ((1, 16, 9), 4), ((3, 2, 4, 4), 9), ((2, 96, 13, 13), 5)):
        xs = (rng.standard_normal(shape) * 2).astype(np.float32)
        got = host(ops.lrn(rt, dev(xs, TD[dt]), size, alpha=1e-2, beta=0.75, bias=2.0))
        want = R.lrn(R.round_to(xs, dt), size, 1e-2, 0.75, 2.0)
        assert np.allclose(got, want, rtol=max(tol, 2e-6) * 4, atol=max(tol, 2e-6) * 4)


def test_conv_packed_weight_cache(rt):
    """Constant weights are re-packed ONCE (runtime-owned image keyed by pointer / shape), reused by eager calls and by a
    captured graph (packed on a side stream while the runtime stream records: no pack node in the graph), and dropped
    when the host copies new values over the weight tensor."""
    from infinitensor_amd import RocmRuntime

    r = RocmRuntime(0)
    rng = np.random.default_rng(3)
    x = dev(rng.standard_normal((2, 64, 14, 14)).astype(np.float32), torch.float16)
    w1 = (rng.standard_normal((128, 64, 3, 3)) / 24).astype(np.float16)
    w2 = (rng.standard_normal((128, 64, 3, 3)) / 24).astype(np.float16)
    w = torch.from_numpy(w1).cuda()
    torch.cuda.synchronize()
    y_plain = ops.conv2d(r, x, w, 1, 1)
    r.sync()
    assert ops.weight_cache_info(r)["entries"] == 0
    ops.set_conv_const_weights(r, True)
    try:
        ya = ops.conv2d(r, x, w, 1, 1)
        yb = ops.conv2d(r, x, w, 1, 1)
        r.sync()
        info = ops.weight_cache_info(r)
        assert info["entries"] == 1 and info["bytes"] >= w.numel() * 2
        assert torch.equal(ya, y_plain) and torch.equal(yb, y_plain)
        # a second weight tensor is packed INSIDE a capture (cold cache): the graph replays against the cached image
        wb = torch.from_numpy(w2).cuda()
        yc = torch.empty_like(y_plain)
        torch.cuda.synchronize()
        r.begin_capture()
        ops.conv2d(r, x, wb, 1, 1, out=yc)
        g = r.end_capture()
        assert ops.weight_cache_info(r)["entries"] == 2
        for _ in range(2):
            yc.zero_()
            torch.cuda.synchronize()
            r.launch_graph(g)
            r.sync()
            want = R.conv2d(host(x).astype(np.float64), w2.astype(np.float64), 1, 1, 1, 1, 1, 1)
            assert np.allclose(host(yc), want, rtol=2e-3, atol=2e-3)
        # new VALUES in the first weight tensor through the runtime's copy: its image is dropped, the next call re-packs
        e0 = ops.weight_cache_info(r)["epoch"]
        src = np.ascontiguousarray(w2)
        r.copy_from_cpu(w.data_ptr(), src.ctypes.data, src.nbytes)
        info = ops.weight_cache_info(r)
        assert info["entries"] == 1 and info["epoch"] == e0 + 1
        yd = ops.conv2d(r, x, w, 1, 1)
        r.sync()
        assert ops.weight_cache_info(r)["entries"] == 2
        assert np.allclose(host(yd), want, rtol=2e-3, atol=2e-3) and not torch.equal(yd, y_plain)
    finally:
        ops.set_conv_const_weights(r, False)


def test_conv_s1_zero_padding_is_exact(rt):
    """An all-ones 3x3 over an all-ones image counts the taps inside the image: 4 / 6 / 9 exactly."""
    x = torch.ones((2, 64, 6, 7), dtype=torch.float16).cuda()
    w = torch.ones((3, 64, 3, 3), dtype=torch.float16).cuda()
    ops.set_conv_variant(rt, 2)
    try:
        y = host(ops.conv2d(rt, x, w, 1, 1))
    finally:
        ops.set_conv_variant(rt, -1)
    cnt = R.conv2d(np.ones((1, 1, 6, 7)), np.ones((1, 1, 3, 3)), 1, 1, 1, 1, 1, 1)[0, 0]
    assert np.array_equal(y, np.broadcast_to(cnt * 64, y.shape))


def test_conv_fused_bias_relu_equals_unfused_chain(rt):
    """Fusion must be invisible: conv(bias, relu) == relu(conv + bias) computed op by op."""
    rng = np.random.default_rng(4)
    x = dev(rng.standard_normal((2, 32, 20, 20)).astype(np.float32), torch.float16)
    w = dev((rng.standard_normal((64, 32, 3, 3)) / 17).astype(np.float32), torch.float16)
    b = dev(rng.standard_normal((64,)).astype(np.float32), torch.float16)
    fused = ops.conv2d(rt, x, w, 1, 1, bias=b, act=1)
    y = ops.conv2d(rt, x, w, 1, 1)
    y = ops.binary(rt, "add", y, b.reshape(1, 64, 1, 1).contiguous())
    y = ops.unary(rt, "relu", y)
    # the unfused chain rounds to fp16 after the conv and after the add; the fused kernel rounds once
    assert np.allclose(host(fused), host(y), rtol=2e-3, atol=2e-3)


def _reduce_cases():
    import json
    from conftest import REPO

    recs = json.loads((REPO / "tests/golden/kats.json").read_text())[CU + "test_cuda_reduce.cc"]
    cases, i = [], 0
    while i < len(recs):
        shape, x = recs[i]["values"], recs[i + 1]["values"]
        if recs[i + 2]["kind"] == "int":
            axes, want = recs[i + 2]["values"], recs[i + 3]["values"]
            i += 4
        else:
            axes, want = None, recs[i + 2]["values"]
            i += 3
        cases.append((shape, x, axes, want))
    return cases


def test_reduce_reference_kats(rt):
    """test_cuda_reduce.cc:42-75 (keepdims alternates true/false; data identical)."""
    for n, (shape, x, axes, want) in enumerate(_reduce_cases()):
        kind = "mean" if n < 4 else "sum"
        y = ops.reduce(rt, kind, dev(np.array(x, dtype=np.float32).reshape(shape)), axes, keepdims=bool(n % 2 == 0))
        assert R.equal_data(host(y).ravel(), np.array(want, dtype=np.float64), 1e-6), (n, kind)


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("shape,axes", [((4, 512, 768), [2]), ((3, 5, 7, 9), [1, 3]), ((6, 40), [0]), ((2, 3, 4, 5), None),
                                        ((8, 2048, 7, 7), [2, 3]), ((5, 1, 6), [0, 1]), ((3, 1000), [-1]),
                                        # short trailing rows (< 64 elements): the LDS-staged kernel — ragged last block, even row length,
                                        # rows that do not start on 16-byte boundaries
                                        ((3, 431, 4, 4), [2, 3]), ((300, 63), [1]), ((1293, 2), [1])])
def test_reduce_vs_oracle(rt, shape, axes, dt):
    rng = np.random.default_rng(21)
    x = rng.standard_normal(shape).astype(np.float32)
    for kind in ("sum", "mean"):
        y = ops.reduce(rt, kind, dev(x, TD[dt]), axes, keepdims=True)
        want = R.reduce(kind, R.round_to(x, dt), axes or [], True)
        tol = {"f32": 1e-4, "f16": 2e-3, "bf16": 1.6e-2}[dt]
        assert y.shape == want.shape
        assert np.allclose(host(y), want, rtol=tol, atol=tol * max(1.0, np.sqrt(x.size / want.size)))


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("rows,n", [(5, 7), (301, 49), (3, 63), (9, 5)])  # odd rows * n: the span ends on a half word
def test_reduce_short_rows_inf_stays_in_its_own_row(rt, rows, n, dt):
    """16-bit short-row sums read two elements per LDS word; the word a row shares with its neighbour must contribute the row's own
    half only — a neighbour's +-Inf / NaN (an attention mask, an f16 overflow) may not reach this row's sum (0 x Inf = NaN), and the
    stale half word behind an odd span may not reach the last row's."""
    rng = np.random.default_rng(rows * 100 + n)
    x = rng.standard_normal((rows, n)).astype(np.float32)
    x[1, 0] = np.inf      # first element of row 1: shares its word with row 0's last element when n is odd
    x[1, n - 1] = -np.inf if rows > 3 else np.inf  # last element of row 1: shares with row 2's first
    if rows > 4:
        x[4, n // 2] = np.nan
    for kind in ("sum", "mean"):
        # run twice behind a launch that leaves Inf patterns in LDS-sized scratch: the stale-half case must not depend on luck
        for _ in range(2):
            junk = ops.reduce(rt, "sum", dev(np.full((rows + 1, n + 1), np.inf, dtype=np.float32), TD[dt]), [1], keepdims=True)
            y = host(ops.reduce(rt, kind, dev(x, TD[dt]), [1], keepdims=True)).ravel()
            del junk
            want = R.reduce(kind, R.round_to(x, dt), [1], True).ravel()
            fin = np.isfinite(want)
            assert not fin[1] and (rows <= 4 or not fin[4])
            assert np.array_equal(np.isnan(y), np.isnan(want)) and np.array_equal(np.isinf(y), np.isinf(want)), (kind, y, want)
            assert np.array_equal(np.sign(y[np.isinf(y)]), np.sign(want[np.isinf(want)]))
            tol = {"f16": 2e-3, "bf16": 1.6e-2}[dt]
            assert np.allclose(y[fin], want[fin], rtol=tol, atol=tol * np.sqrt(n)), kind


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_global_avgpool_small_planes_inf_stays_in_its_own_plane(rt, dt):
    """The same shared-word rule in the small-plane global average pool (7 x 7 planes are 49 elements: every other plane starts on an
    odd element); 3 x 5 x 7 x 7 = an odd element count, so the last plane's last word is half foreign."""
    rng = np.random.default_rng(77)
    x = rng.standard_normal((3, 5, 7, 7)).astype(np.float32)
    x[0, 1, 0, 0] = np.inf
    x[0, 1, 6, 6] = np.inf
    x[2, 3, 3, 3] = -np.inf
    for _ in range(2):
        y = host(ops.avg_pool(rt, dev(x, TD[dt]), 7, 7, 1, 1, 0, 0, 1, 1, 0)).ravel()
        want = R.pool2d(R.round_to(x, dt), "avg", 7, 7, 1, 1, 0, 0, 1, 1, 0).ravel()
        fin = np.isfinite(want)
        assert fin.sum() == 13
        assert np.array_equal(np.isinf(y), np.isinf(want)) and not np.isnan(y).any(), (y, want)
        assert np.array_equal(np.sign(y[~fin]), np.sign(want[~fin]))
        tol = {"f16": 2e-3, "bf16": 1.6e-2}[dt]
        assert np.allclose(y[fin], want[fin], rtol=tol, atol=tol)


def test_batchnorm_reference_kat(rt):
    f = CU + "test_cuda_batch_norm.cc"
    y = ops.batch_norm(rt, dev(R.incremental((1, 3, 2, 2))), dev(kat(f, 25, "float").astype(np.float32)),
                       dev(kat(f, 26, "float").astype(np.float32)), dev(R.ones((3,))), dev(np.zeros(3, np.float32)), 0.0)
    assert R.equal_data(host(y).ravel(), kat(f, 51, "float"), 2e-6)


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("shape", [(4, 64, 56, 56), (2, 3, 5, 7), (3, 10), (2, 7, 9)])
def test_batchnorm_vs_oracle(rt, shape, dt):
    rng = np.random.default_rng(23)
    c = shape[1]
    x = rng.standard_normal(shape).astype(np.float32)
    m, v = rng.standard_normal(c).astype(np.float32), rng.uniform(0.5, 2, c).astype(np.float32)
    s, b = rng.standard_normal(c).astype(np.float32), rng.standard_normal(c).astype(np.float32)
    y = ops.batch_norm(rt, dev(x, TD[dt]), dev(m), dev(v), dev(s), dev(b), 1e-5)
    want = R.batch_norm(R.round_to(x, dt), m, v, s, b, 1e-5)
    tol = {"f32": 1e-4, "f16": 2e-3, "bf16": 1.6e-2}[dt]
    assert np.allclose(host(y), want, rtol=tol, atol=tol)


def test_pooling_reference_kats(rt):
    f = CU + "test_cuda_pooling.cc"
    x = dev(R.incremental((1, 2, 5, 5)))
    assert R.equal_data(host(ops.max_pool(rt, x, 3, 3, 1, 1, 1, 1, 2, 2)).ravel(), kat(f, 48, "float"), 1e-6)
    assert R.equal_data(host(ops.avg_pool(rt, x, 3, 3, 1, 1, 1, 1, 2, 2)).ravel(), kat(f, 55, "float"), 2e-6)


@pytest.mark.parametrize("dt", ["f32", "f16"])
@pytest.mark.parametrize("cfg", [((2, 64, 112, 112), 3, 3, 1, 1, 1, 1, 2, 2, 0), ((2, 2048, 7, 7), 7, 7, 1, 1, 0, 0, 1, 1, 0),
                                 ((1, 3, 10, 9), 2, 3, 2, 1, 1, 0, 1, 2, 1), ((2, 5, 8), 1, 3, 1, 1, 0, 1, 1, 2, 0),
                                 ((3, 5, 9, 16), 3, 3, 1, 1, 1, 1, 2, 2, 0),  # odd height on the specialised 3x3/2 max-pool kernel
                                 ((3, 100, 7, 7), 7, 7, 1, 1, 0, 0, 1, 1, 0),  # global pools: 300 planes = a ragged last slab of
                                 ((1, 513, 5, 3), 5, 3, 1, 1, 0, 0, 1, 1, 0)])  # the small-plane kernel; odd plane sizes
def test_pooling_vs_oracle(rt, cfg, dt):
    shape, kh, kw, dh, dw, ph, pw, sh, sw, ceil = cfg
    rng = np.random.default_rng(29)
    x = rng.standard_normal(shape).astype(np.float32)
    x4 = x if x.ndim == 4 else x[:, :, None, :]
    for kind, fn in (("max", ops.max_pool), ("avg", ops.avg_pool)):
        y = fn(rt, dev(x, TD[dt]), kh, kw, dh, dw, ph, pw, sh, sw, ceil)
        want = R.pool2d(R.round_to(x4, dt), kind, kh, kw, dh, dw, ph, pw, sh, sw, ceil)
        if x.ndim == 3:
            want = want[:, :, 0, :]
        assert tuple(y.shape) == want.shape, kind
        tol = {"f32": 1e-5, "f16": 2e-3}[dt]
        assert np.allclose(host(y), want, rtol=tol, atol=tol), kind


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("shape", [(1, 1, 2, 8), (1, 2, 3, 8), (2, 3, 5, 24), (1, 4, 16, 16), (2, 2, 31, 40), (1, 3, 64, 112), (5, 7, 12, 8)])
def test_maxpool_3x3_stride2_kernel_shapes(rt, shape, relu, dt):
    """The specialised MaxPool 3 x 3 / 2 / pad 1 kernel (two output rows per thread since round 5; reference pooling.cc:6-95): one and
    two output rows, odd heights (the last thread's second row does not exist, the last input row is missing), one and several
    4-output groups per row, with and without the fused ReLU — bit-exact (a maximum of stored values)."""
    rng = np.random.default_rng(sum(shape))
    x = R.round_to(rng.standard_normal(shape).astype(np.float32), dt)
    xd = dev(x, TD[dt])
    y = ops.max_pool(rt, xd, 3, 3, 1, 1, 1, 1, 2, 2, 0)
    want = R.pool2d(x, "max", 3, 3, 1, 1, 1, 1, 2, 2, 0)
    if relu:
        import ctypes as C
        from infinitensor_amd._lib import check, lib
        n, c, h, w = shape
        y = torch.empty(want.shape, device="cuda", dtype=TD[dt])
        check(lib().infini_rocm_pool2d_relu(rt.handle, 0, ops.dtype_of(xd), C.c_void_p(xd.data_ptr()), C.c_void_p(y.data_ptr()), n, c, h, w, 3, 3, 1, 1, 1, 1,
                                            2, 2, 0, 1))
        want = np.maximum(want, 0)
    assert tuple(y.shape) == want.shape
    assert np.array_equal(host(y), want.astype(np.float32))


CONVT = [
    # n, f, h, w, cg, r, s, ph, pw, sh, sw, dh, dw, oph, opw, groups
    (2, 8, 5, 6, 4, 3, 3, 1, 1, 2, 2, 1, 1, 1, 1, 1),   # the usual 2x up-sampling deconv
    (1, 16, 4, 4, 8, 4, 4, 1, 1, 2, 2, 1, 1, 0, 0, 1),  # DCGAN 4x4 / 2
    (2, 6, 7, 5, 2, 3, 2, 0, 1, 1, 3, 2, 1, 0, 2, 3),   # groups, dilation, asymmetric stride + output padding
    (1, 4, 3, 3, 5, 1, 1, 0, 0, 1, 1, 1, 1, 0, 0, 1),   # 1x1
]


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("cfg", CONVT)
def test_conv_transpose_vs_oracle(rt, cfg, dt):
    n, f, h, w, cg, r, s, ph, pw, sh, sw, dh, dw, oph, opw, g = cfg
    rng = np.random.default_rng(abs(hash(cfg)) % 2 ** 32)
    x = rng.standard_normal((n, f, h, w)).astype(np.float32)
    wt = (rng.standard_normal((f, cg, r, s)) / np.sqrt(f * r * s / g)).astype(np.float32)
    y = ops.conv_transpose2d(rt, dev(x, TD[dt]), dev(wt, TD[dt]), ph, pw, sh, sw, dh, dw, oph, opw, g)
    want = R.conv_transpose2d(R.round_to(x, dt), R.round_to(wt, dt), ph, pw, sh, sw, dh, dw, oph, opw, g)
    assert tuple(y.shape) == want.shape
    tol = {"f32": 1e-5, "f16": 2e-3, "bf16": 1.6e-2}[dt]
    assert np.allclose(host(y), want, rtol=tol, atol=tol)


def test_conv_transpose_reference_kats(rt):
    """test_cuda_conv_transposed_2d.cc:85-91 and :101-135."""
    CT = CU + "test_cuda_conv_transposed_2d.cc"
    y = ops.conv_transpose2d(rt, dev(R.incremental((1, 1, 2, 2))), dev(R.incremental((1, 1, 4, 4))))
    assert R.equal_data(host(y).ravel(), kat(CT, 87, "float"), 1e-6)
    y = ops.conv_transpose2d(rt, dev(R.incremental((1, 2, 3, 3))), dev(R.incremental((2, 2, 3, 3))))
    assert R.equal_data(host(y).ravel(), kat(CT, 129, "float"), 1e-6)


@pytest.mark.parametrize("dt", ["f32", "f16"])
@pytest.mark.parametrize("cfg", [(2, 64, 14, 14, 128, 1, 1, 0, 1), (2, 32, 9, 9, 40, 3, 3, 1, 1), (1, 64, 8, 8, 256, 1, 1, 0, 2), (2, 3, 16, 16, 8, 7, 7, 3, 2),
                                 # the LDS-staged epilogue with a residual: conv_pw (C <= 128) and conv_s1, filter counts that
                                 # leave partial tiles, planes of 196 pixels (a 4-pixel last run per row), several images
                                 (3, 128, 14, 14, 200, 1, 1, 0, 1), (2, 64, 28, 28, 256, 1, 1, 0, 1), (3, 256, 14, 14, 320, 3, 3, 1, 1),
                                 (2, 192, 6, 6, 72, 1, 1, 0, 1)])
def test_conv_with_residual(rt, cfg, dt):
    """conv2d_res: act(conv + bias + residual) vs the oracle, on the conv_s1 / generic / fp32 paths."""
    n, c, h, w, f, r, s, pad, st = cfg
    rng = np.random.default_rng(abs(hash(cfg)) % 2 ** 32)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = (rng.standard_normal((f, c, r, s)) / np.sqrt(c * r * s)).astype(np.float32)
    b = rng.standard_normal((f,)).astype(np.float32)
    base = R.conv2d(R.round_to(x, dt), R.round_to(wt, dt), pad, pad, st, st, 1, 1)
    res = rng.standard_normal(base.shape).astype(np.float32)
    y = ops.conv2d(rt, dev(x, TD[dt]), dev(wt, TD[dt]), pad, pad, st, st, bias=dev(b, TD[dt]), act=1, residual=dev(res, TD[dt]))
    want = np.maximum(base + R.round_to(b, dt).reshape(1, f, 1, 1) + R.round_to(res, dt), 0)
    tol = {"f32": 1e-4, "f16": 3e-3}[dt]
    assert np.allclose(host(y), want, rtol=tol, atol=tol)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("act", [0, 1])
def test_conv_residual_lds_epilogue_act_and_bf16(rt, dt, act):
    """conv2d_res through the LDS-staged epilogue (even plane, conv_s1 3x3 and conv_pw 1x1): no activation / ReLU after
    the residual add, f16 and bf16; the residual tensor is NOT the output buffer and stays untouched."""
    rng = np.random.default_rng(77 + act)
    for (n, c, h, f, r, pad) in ((3, 128, 14, 144, 1, 0), (2, 64, 12, 136, 3, 1)):
        x = rng.standard_normal((n, c, h, h)).astype(np.float32)
        wt = (rng.standard_normal((f, c, r, r)) / np.sqrt(c * r * r)).astype(np.float32)
        b = rng.standard_normal((f,)).astype(np.float32)
        base = R.conv2d(R.round_to(x, dt), R.round_to(wt, dt), pad, pad, 1, 1, 1, 1)
        res = rng.standard_normal(base.shape).astype(np.float32)
        dres = dev(res, TD[dt])
        keep = dres.clone()
        y = ops.conv2d(rt, dev(x, TD[dt]), dev(wt, TD[dt]), pad, pad, 1, 1, bias=dev(b, TD[dt]), act=act, residual=dres)
        want = base + R.round_to(b, dt).reshape(1, f, 1, 1) + R.round_to(res, dt)
        if act:
            want = np.maximum(want, 0)
        tol = {"f16": 3e-3, "bf16": 2.4e-2}[dt]
        assert np.allclose(host(y), want, rtol=tol, atol=tol), (dt, act, c, np.abs(host(y) - want).max())
        assert torch.equal(dres, keep)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("cfg", [(3, 64, 14, 14, 256, 3), (5, 96, 7, 7, 130, 3), (2, 32, 28, 28, 128, 3), (9, 64, 9, 5, 200, 3), (2, 64, 12, 12, 128, 5)])
def test_conv_patch_wide_form(rt, cfg, dt):
    """The 8-wave 128 f x 256 slots form of the LDS-resident-patch kernel (conv variant 6; three weight stages, one workgroup per
    CU): unit-stride "same" R x S layers with ragged planes, filter counts that are not multiples of 128, a slot count that is not
    a multiple of 256 — against the oracle and against the 4-wave form (variant 2), with bias + ReLU."""
    n, c, h, w, f, k = cfg
    rng = np.random.default_rng(abs(hash(cfg)) % 2 ** 32)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = (rng.standard_normal((f, c, k, k)) / np.sqrt(c * k * k)).astype(np.float32)
    b = rng.standard_normal((f,)).astype(np.float32)
    xd, wd, bd = dev(x, TD[dt]), dev(wt, TD[dt]), dev(b, TD[dt])
    try:
        ops.set_conv_variant(rt, 6)
        y = ops.conv2d(rt, xd, wd, k // 2, k // 2, 1, 1, bias=bd, act=1)
        ops.set_conv_variant(rt, 2)
        y2 = ops.conv2d(rt, xd, wd, k // 2, k // 2, 1, 1, bias=bd, act=1)
    finally:
        ops.set_conv_variant(rt, -1)
    want = np.maximum(R.conv2d(R.round_to(x, dt), R.round_to(wt, dt), k // 2, k // 2, 1, 1, 1, 1) + R.round_to(b, dt).reshape(1, f, 1, 1), 0)
    tol = {"f16": 3e-3, "bf16": 2.4e-2}[dt]
    assert np.allclose(host(y), want, rtol=tol, atol=tol), np.abs(host(y) - want).max()
    assert np.allclose(host(y), host(y2), rtol=tol, atol=tol)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("cfg", [(2, 64, 8, 8, 64, 3), (3, 32, 16, 16, 48, 3), (5, 64, 12, 12, 64, 3), (70, 64, 8, 8, 64, 3), (1, 64, 56, 56, 64, 3),
                                 (3, 64, 4, 6, 17, 3)])
def test_conv_resident_weights_kernel(rt, cfg, dt):
    """F <= 64, C <= 64 unit-stride "same" layers on planes that are multiples of 8 pixels: the whole weight tensor resident in LDS,
    persistent workgroups over 256-slot tiles (conv_resident_kernel). One tile, several tiles per workgroup (70 x 64 slots = 18
    tiles... on 256 CUs one each; 1 x 3136 = 13 tiles), a ragged last tile, C = 32 (one channel block, odd step count), ragged
    filter counts; bias + ReLU; against the oracle and the tap-shifted kernel (variant 4)."""
    n, c, h, w, f, k = cfg
    rng = np.random.default_rng(abs(hash(cfg)) % 2 ** 32)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = (rng.standard_normal((f, c, k, k)) / np.sqrt(c * k * k)).astype(np.float32)
    b = rng.standard_normal((f,)).astype(np.float32)
    xd, wd, bd = dev(x, TD[dt]), dev(wt, TD[dt]), dev(b, TD[dt])
    try:
        y = ops.conv2d(rt, xd, wd, k // 2, k // 2, 1, 1, bias=bd, act=1)
        assert ops.conv_last_route(rt) == "resident"
        ops.set_conv_variant(rt, 4)
        y2 = ops.conv2d(rt, xd, wd, k // 2, k // 2, 1, 1, bias=bd, act=1)
        assert ops.conv_last_route(rt) == "tap_shifted"
    finally:
        ops.set_conv_variant(rt, -1)
    want = np.maximum(R.conv2d(R.round_to(x, dt), R.round_to(wt, dt), k // 2, k // 2, 1, 1, 1, 1) + R.round_to(b, dt).reshape(1, f, 1, 1), 0)
    tol = {"f16": 3e-3, "bf16": 2.4e-2}[dt]
    assert np.allclose(host(y), want, rtol=tol, atol=tol), np.abs(host(y) - want).max()
    assert np.allclose(host(y), host(y2), rtol=tol, atol=tol)


PW_GEMM = [  # (n, c, h, w, f): pointwise layers for the conv mode of the persistent GEMM (variant 5)
    (3, 64, 8, 8, 256),      # plane of 64 pixels: a 256-slot tile spans four images
    (5, 128, 14, 14, 512),   # 196 pixels: 4-pixel ragged run at every plane end, tiles span images
    (9, 192, 7, 7, 320),     # 49 pixels: 1-pixel ragged run, filters not a multiple of 256 (ragged M), K = 3 tiles
    (2, 256, 28, 28, 256),   # 784 pixels: whole tiles inside a plane
    (1, 64, 3, 3, 64),       # 9 pixels (below the kernel's minimum of 8 usable... 9 >= 8: one run + one pixel), tiny
    (2, 64, 5, 4, 300),      # 20 pixels: 4-pixel tail
    (4, 64, 6, 5, 260),      # 30 pixels: 6-pixel tail (4 + 2)
    (2, 64, 5, 3, 256),      # 15 pixels: 7-pixel tail (4 + 2 + 1)
]


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("mode", ["plain", "bias_relu", "bias_res_relu", "res"])
@pytest.mark.parametrize("cfg", PW_GEMM)
def test_conv_pointwise_gemm_mode(rt, cfg, dt, mode):
    """Unit-stride 1 x 1 convolutions as ONE GEMM over pixel slots on the persistent 256-row kernels (gemm256p_kernel.h, CONV;
    conv variant 5, the default for >= 256 filters): planes that are and are not multiples of 8 pixels (ragged last run of every
    plane: 4 + 2 + 1 stores), tiles spanning images, ragged filter counts, with per-filter bias / residual / ReLU in the
    epilogue — against the oracle and against the tap-shifted kernel (variant 2)."""
    n, c, h, w, f = cfg
    rng = np.random.default_rng(abs(hash(cfg)) % 2 ** 32)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = (rng.standard_normal((f, c, 1, 1)) / np.sqrt(c)).astype(np.float32)
    b = rng.standard_normal((f,)).astype(np.float32) if "bias" in mode else None
    res = rng.standard_normal((n, f, h, w)).astype(np.float32) if "res" in mode else None
    act = 1 if "relu" in mode else 0
    xd, wd = dev_slack(x, TD[dt]), dev(wt, TD[dt])
    bd = dev(b, TD[dt]) if b is not None else None
    rd = dev(res, TD[dt]) if res is not None else None
    keep = rd.clone() if rd is not None else None
    guard = torch.full((n, f, h, w), 7.0, device="cuda", dtype=TD[dt])  # the output buffer, pre-filled: every element must be written
    try:
        ops.set_conv_variant(rt, 5)
        y = ops.conv2d(rt, xd, wd, 0, 0, 1, 1, bias=bd, act=act, residual=rd, out=guard)
        assert ops.conv_last_route(rt) == "pixel_gemm"  # not a silent fall-back to the kernel it is compared with
        ops.set_conv_variant(rt, 2)
        y2 = ops.conv2d(rt, xd, wd, 0, 0, 1, 1, bias=bd, act=act, residual=rd)
    finally:
        ops.set_conv_variant(rt, -1)
    want = R.conv2d(R.round_to(x, dt), R.round_to(wt, dt), 0, 0, 1, 1, 1, 1)
    if b is not None:
        want = want + R.round_to(b, dt).reshape(1, f, 1, 1)
    if res is not None:
        want = want + R.round_to(res, dt)
        assert torch.equal(rd, keep)
    if act:
        want = np.maximum(want, 0)
    tol = {"f16": 3e-3, "bf16": 2.4e-2}[dt]
    assert np.allclose(host(y), want, rtol=tol, atol=tol), np.abs(host(y) - want).max()
    assert np.allclose(host(y), host(y2), rtol=tol, atol=tol)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("cfg", [(4, 256, 28, 28, 512, 2), (8, 128, 15, 13, 256, 2), (2, 64, 30, 30, 256, 3), (2, 64, 56, 56, 128, 2),
                                 (6, 64, 15, 14, 256, 2), (3, 128, 14, 14, 256, 2)])
def test_conv_strided_pointwise_goes_through_the_phase_split_into_the_gemm_mode(rt, cfg, dt):
    """ResNet's down-sampling 1 x 1 / 2 layers: the phase split leaves ONE dense [n][c][oh][ow] plane set, which then is a
    unit-stride pointwise layer for the conv mode of the persistent GEMM (conv_s1.hip, launch_conv_s1). Odd input sizes (the
    sampled grid ends before the input does) and stride 3 included; against the oracle and the tap-shifted kernel."""
    n, c, h, w, f, st = cfg
    rng = np.random.default_rng(abs(hash(cfg)) % 2 ** 32)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = (rng.standard_normal((f, c, 1, 1)) / np.sqrt(c)).astype(np.float32)
    b = rng.standard_normal((f,)).astype(np.float32)
    xd, wd, bd = dev(x, TD[dt]), dev(wt, TD[dt]), dev(b, TD[dt])  # (the GEMM reads the phase planes in the workspace, not x)
    try:
        ops.set_conv_variant(rt, 5)
        y = ops.conv2d(rt, xd, wd, 0, 0, st, st, bias=bd, act=0)
        assert ops.conv_last_route(rt) == "pixel_gemm"
        ops.set_conv_variant(rt, 2)
        y2 = ops.conv2d(rt, xd, wd, 0, 0, st, st, bias=bd, act=0)
        assert ops.conv_last_route(rt) == "tap_shifted"
    finally:
        ops.set_conv_variant(rt, -1)
    want = R.conv2d(R.round_to(x, dt), R.round_to(wt, dt), 0, 0, st, st, 1, 1) + R.round_to(b, dt).reshape(1, f, 1, 1)
    tol = {"f16": 3e-3, "bf16": 2.4e-2}[dt]
    assert np.allclose(host(y), want, rtol=tol, atol=tol), np.abs(host(y) - want).max()
    assert np.allclose(host(y), host(y2), rtol=tol, atol=tol)


@pytest.mark.parametrize("nt", [2, 3, 4])
@pytest.mark.parametrize("mode", ["bias_relu", "bias_res_relu"])
@pytest.mark.parametrize("cfg", [(5, 128, 14, 14, 512), (3, 64, 8, 8, 256), (9, 192, 7, 7, 320)])
def test_conv_pointwise_gemm_mode_every_tile_width(rt, cfg, mode, nt, monkeypatch):
    """Every instantiation of the conv mode (128 / 192 / 256-column tiles, with and without a residual; whole-run planes — 8 x 8 —
    and ragged ones — 14 x 14, 7 x 7) on the same layers: the cost model alone would pick one width per shape (IROCM_CONV_PW_NT is
    the test hook that forces it)."""
    n, c, h, w, f = cfg
    monkeypatch.setenv("IROCM_CONV_PW_NT", str(nt))
    rng = np.random.default_rng(abs(hash((cfg, nt))) % 2 ** 32)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = (rng.standard_normal((f, c, 1, 1)) / np.sqrt(c)).astype(np.float32)
    b = rng.standard_normal((f,)).astype(np.float32)
    res = rng.standard_normal((n, f, h, w)).astype(np.float32) if "res" in mode else None
    xd, wd, bd = dev_slack(x, torch.float16), dev(wt, torch.float16), dev(b, torch.float16)
    rd = dev(res, torch.float16) if res is not None else None
    try:
        ops.set_conv_variant(rt, 5)
        y = ops.conv2d(rt, xd, wd, 0, 0, 1, 1, bias=bd, act=1, residual=rd)
        assert ops.conv_last_route(rt) == "pixel_gemm"
    finally:
        ops.set_conv_variant(rt, -1)
    want = R.conv2d(R.round_to(x, "f16"), R.round_to(wt, "f16"), 0, 0, 1, 1, 1, 1) + R.round_to(b, "f16").reshape(1, f, 1, 1)
    if res is not None:
        want = want + R.round_to(res, "f16")
    want = np.maximum(want, 0)
    assert np.allclose(host(y), want, rtol=3e-3, atol=3e-3), np.abs(host(y) - want).max()


def test_conv_pointwise_gemm_mode_does_not_write_outside_its_output(rt):
    """The NCHW store of the conv mode ends every plane with a ragged run: the bytes right behind the output tensor (and the
    residual's) must stay untouched."""
    n, c, h, f = 3, 64, 7, 256
    buf = torch.full((n * f * h * h + 64,), 3.0, device="cuda", dtype=torch.float16)
    out = buf[: n * f * h * h].view(n, f, h, h)
    x = dev_slack(np.random.default_rng(3).standard_normal((n, c, h, h)).astype(np.float32), torch.float16)
    w = (torch.randn(f, c, 1, 1, device="cuda") / 8).half()
    try:
        ops.set_conv_variant(rt, 5)
        ops.conv2d(rt, x, w, 0, 0, 1, 1, out=out)
        assert ops.conv_last_route(rt) == "pixel_gemm"
    finally:
        ops.set_conv_variant(rt, -1)
    rt.sync()
    assert torch.all(buf[n * f * h * h:] == 3.0).item()
    ref = torch.einsum("fc,nchw->nfhw", w.view(f, c).float(), x.float())
    assert torch.allclose(out.float(), ref, rtol=3e-3, atol=3e-3)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("cfg", [(2, 32, 32), (3, 64, 40), (1, 224, 224), (2, 70, 120), (1, 23, 16), (2, 9, 8)])
def test_conv_stem_pool_fused_vs_oracle(rt, cfg, dt):
    """infini_rocm_conv2d_pool: Conv(7 x 7 / 2 / 3, C = 3 -> F = 64) + bias + ReLU + MaxPool(3 x 3 / 2 / 1) as one launch
    (csrc/conv_stem.hip) against the oracle's conv2d -> + bias -> round -> relu -> pool2d, and against the library's own
    unfused chain (conv2d with the bias / ReLU epilogue, then max_pool): whole planes, ragged tiles in both directions (the
    workgroup tile is 2 x 28 pooled outputs), odd conv / pool extents, an image smaller than one tile."""
    n, h, w = cfg
    rng = np.random.default_rng(abs(hash(cfg)) % 2 ** 32)
    x = rng.random((n, 3, h, w)).astype(np.float32) * 2 - 0.5
    wt = (rng.standard_normal((64, 3, 7, 7)) * np.sqrt(2 / 147)).astype(np.float32)
    b = (rng.standard_normal(64) * 0.2).astype(np.float32)
    xd, wd, bd = dev(x, TD[dt]), dev(wt, TD[dt]), dev(b, TD[dt])
    y = ops.conv2d_pool(rt, xd, wd, bd, 3, 3, 2, 2, 3, 2, 1)
    assert ops.conv_last_route(rt) == "stem_pool"
    conv = R.conv2d(R.round_to(x, dt), R.round_to(wt, dt), 3, 3, 2, 2, 1, 1) + R.round_to(b, dt).reshape(1, 64, 1, 1)
    want = R.pool2d(np.maximum(R.round_to(conv, dt), 0), "max", 3, 3, 1, 1, 1, 1, 2, 2, 0)
    assert tuple(y.shape) == want.shape
    tol = {"f16": 3e-3, "bf16": 2.4e-2}[dt]
    assert np.allclose(host(y), want, rtol=tol, atol=tol), np.abs(host(y) - want).max()
    chain = ops.max_pool(rt, ops.conv2d(rt, xd, wd, 3, 3, 2, 2, bias=bd, act=1), 3, 3, 1, 1, 1, 1, 2, 2, 0)
    assert np.allclose(host(y), host(chain), rtol=tol, atol=tol)


def test_conv_stem_pool_declines_what_it_does_not_serve(rt):
    import torch

    x = torch.zeros((1, 3, 32, 32), dtype=torch.float16, device="cuda")
    w5 = torch.zeros((64, 3, 5, 5), dtype=torch.float16, device="cuda")
    with pytest.raises(RuntimeError):
        ops.conv2d_pool(rt, x, w5, None, 2, 2, 2, 2, 3, 2, 1)
    w7 = torch.zeros((64, 3, 7, 7), dtype=torch.float16, device="cuda")
    with pytest.raises(RuntimeError):
        ops.conv2d_pool(rt, x, w7, None, 3, 3, 2, 2, 2, 2, 0)  # another pooling window
    with pytest.raises(RuntimeError):
        ops.conv2d_pool(rt, x.float(), w7.float(), None, 3, 3, 2, 2, 3, 2, 1)  # fp32 keeps the separate kernels


def dev_slack2(a, dtype, fill=float("nan"), spare=512):
    """`a` on the device in the MIDDLE of a block whose `spare` elements on both sides hold `fill`: the tap mode (3 x 3 layers as one
    GEMM) fetches up to one row + one pixel in front of and behind the tensor and must mask all of it away — NaNs would survive."""
    a = np.ascontiguousarray(a)
    buf = torch.full((a.size + 2 * spare,), fill, device="cuda", dtype=dtype)
    return buf[spare: spare + a.size].view(a.shape).copy_(torch.from_numpy(a))


TAP_CFGS = [
    # n, c, h, w, f, stride
    (3, 64, 14, 14, 256, 1),    # ragged plane (196 = 24.5 runs), tiles span images
    (2, 128, 7, 7, 512, 1),     # 49-pixel planes, two filter tiles
    (2, 64, 16, 24, 256, 1),    # whole-run plane, non-square
    (1, 64, 56, 56, 256, 1),    # long rows (front slack 114 bytes)
    (5, 192, 9, 11, 320, 1),    # three channel blocks, ragged filters, odd rows
    (3, 64, 28, 28, 256, 2),    # stride 2: phase planes 14 x 14
    (2, 128, 14, 14, 512, 2),   # stride 2: 7 x 7 output
    (2, 64, 15, 13, 256, 2),    # stride 2 on odd input extents (the last input row / column is read by r = 1 only)
    (2, 64, 30, 18, 300, 2),
]


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("mode", ["plain", "bias_relu"])
@pytest.mark.parametrize("cfg", TAP_CFGS)
def test_conv3x3_tap_gemm_mode(rt, cfg, mode, dt):
    """Round 5: 3 x 3 / pad 1 layers of stride 1 and 2 as ONE GEMM with K = 9 C on the persistent 256-row kernels (conv variant 7,
    route "tap_gemm"; csrc/gemm256p_kernel.h CONV = 3). A tap moves the pointwise tile's 16-byte runs, so the runs at a row's /
    image's / plane's edge drag in neighbours (or the NaN-filled slack around the tensor): the per-lane masks must remove exactly
    those. Against the oracle (reference semantics: src/kernels/cuda/conv.cc:57-168) and against the tap-shifted kernel."""
    n, c, h, w, f, st = cfg
    rng = np.random.default_rng(abs(hash(cfg)) % 2 ** 32)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = (rng.standard_normal((f, c, 3, 3)) / np.sqrt(9 * c)).astype(np.float32)
    b = rng.standard_normal((f,)).astype(np.float32) if "bias" in mode else None
    act = 1 if "relu" in mode else 0
    xd, wd = dev_slack2(x, TD[dt]), dev(wt, TD[dt])
    bd = dev(b, TD[dt]) if b is not None else None
    oh, ow = (h + st - 1) // st, (w + st - 1) // st
    guard = torch.full((n, f, oh, ow), 7.0, device="cuda", dtype=TD[dt])  # every element must be written
    try:
        ops.set_conv_variant(rt, 7)
        y = ops.conv2d(rt, xd, wd, 1, 1, st, st, bias=bd, act=act, out=guard)
        assert ops.conv_last_route(rt).startswith("tap_gemm")  # not a silent fall-back to the kernel it is compared with
        ops.set_conv_variant(rt, 4)
        y2 = ops.conv2d(rt, xd, wd, 1, 1, st, st, bias=bd, act=act)
        assert ops.conv_last_route(rt) == "tap_shifted"
    finally:
        ops.set_conv_variant(rt, -1)
    want = R.conv2d(R.round_to(x, dt), R.round_to(wt, dt), 1, 1, st, st, 1, 1)
    if b is not None:
        want = want + R.round_to(b, dt).reshape(1, f, 1, 1)
    if act:
        want = np.maximum(want, 0)
    tol = {"f16": 3e-3, "bf16": 2.4e-2}[dt]
    got = host(y)
    assert np.isfinite(got).all(), "a masked-away neighbour (NaN slack) leaked into the sum"
    assert np.allclose(got, want, rtol=tol, atol=tol), np.abs(got - want).max()
    assert np.allclose(got, host(y2), rtol=tol, atol=tol)


@pytest.mark.parametrize("nt", [2, 3, 4])
@pytest.mark.parametrize("cfg", [(3, 64, 14, 14, 256, 1), (2, 128, 14, 14, 512, 2), (6, 64, 7, 7, 256, 1)])
def test_conv3x3_tap_gemm_every_tile_width(rt, cfg, nt, monkeypatch):
    """Every instantiation of the tap mode (128 / 192 / 256-column tiles; IROCM_CONV_TAP_NT forces the width the cost model would
    otherwise pick), several tiles per workgroup included (the cursors cross tile boundaries with their tap state)."""
    n, c, h, w, f, st = cfg
    monkeypatch.setenv("IROCM_CONV_TAP_NT", str(nt))
    rng = np.random.default_rng(abs(hash((cfg, nt))) % 2 ** 32)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = (rng.standard_normal((f, c, 3, 3)) / np.sqrt(9 * c)).astype(np.float32)
    b = rng.standard_normal((f,)).astype(np.float32)
    xd, wd, bd = dev_slack2(x, torch.float16), dev(wt, torch.float16), dev(b, torch.float16)
    try:
        ops.set_conv_variant(rt, 7)
        y = ops.conv2d(rt, xd, wd, 1, 1, st, st, bias=bd, act=1)
        assert ops.conv_last_route(rt) == "tap_gemm"  # (a forced width is the unsplit form)
    finally:
        ops.set_conv_variant(rt, -1)
    want = np.maximum(R.conv2d(R.round_to(x, "f16"), R.round_to(wt, "f16"), 1, 1, st, st, 1, 1) + R.round_to(b, "f16").reshape(1, f, 1, 1), 0)
    assert np.allclose(host(y), want, rtol=3e-3, atol=3e-3), np.abs(host(y) - want).max()


def test_conv3x3_tap_gemm_many_tiles_per_workgroup_and_edges(rt):
    """Several slot tiles per workgroup (the A / B cursors run ahead across tile boundaries and reset their tap state there) on a plane
    whose rows are shorter than a 16-byte run (every run straddles rows), plus a tensor that may start its allocation: the launcher
    proves the bytes in front of X readable or falls back — either way the result must be right. Checked on ~6 000 sampled outputs
    (R.conv2d_at: one fp64 dot product each), always including every image's corners and the first / last image."""
    n, c, h, w, f = 1024, 64, 12, 6, 256   # 1024 * 72 = 73 728 slots = 288 tiles of 256 on 256 CUs
    rng = np.random.default_rng(11)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = (rng.standard_normal((f, c, 3, 3)) / np.sqrt(9 * c)).astype(np.float32)
    xd, wd = dev_slack2(x, torch.float16), dev(wt, torch.float16)
    co = np.stack([rng.integers(0, n, 6000), rng.integers(0, f, 6000), rng.integers(0, h, 6000), rng.integers(0, w, 6000)], axis=1)
    fixed = [(i, fi, yy, xx) for i in (0, 1, n // 2, n - 1) for fi in (0, f - 1) for yy in (0, 1, h - 2, h - 1) for xx in (0, 1, w - 2, w - 1)]
    co = np.concatenate([co, np.array(fixed)], axis=0)
    want = R.conv2d_at(R.round_to(x, "f16"), R.round_to(wt, "f16"), co, 1, 1, 1, 1, 1, 1)
    try:
        ops.set_conv_variant(rt, 7)
        y = ops.conv2d(rt, xd, wd, 1, 1, 1, 1)
        assert ops.conv_last_route(rt) == "tap_gemm"
        big = torch.empty(x.size + 64, dtype=torch.float16, device="cuda")  # (likely the start of a fresh caching-allocator segment)
        xe = big[: x.size].view(x.shape).copy_(xd)
        ye = ops.conv2d(rt, xe, wd, 1, 1, 1, 1)
        route_edge = ops.conv_last_route(rt)
    finally:
        ops.set_conv_variant(rt, -1)
    pick = lambda t: t[co[:, 0], co[:, 1], co[:, 2], co[:, 3]].float().cpu().numpy().astype(np.float64)
    ci = torch.from_numpy(co).cuda()
    pick = lambda t: t[ci[:, 0], ci[:, 1], ci[:, 2], ci[:, 3]].float().cpu().numpy().astype(np.float64)
    assert torch.isfinite(y.float()).all().item() and torch.isfinite(ye.float()).all().item()
    assert np.allclose(pick(y), want, rtol=3e-3, atol=3e-3), np.abs(pick(y) - want).max()
    assert route_edge in ("tap_gemm", "tap_shifted")  # (whether `big` starts its segment is the allocator's business; both must be right)
    assert np.allclose(pick(ye), want, rtol=3e-3, atol=3e-3), (route_edge, np.abs(pick(ye) - want).max())


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("cfg", [
    # n, c, h, w, f, stride, split
    (16, 128, 14, 14, 256, 1, 2),    # 13 tiles x 2 slices of 9 K-tiles
    (16, 256, 14, 14, 256, 1, 4),    # 13 tiles x 4 slices of 9 K-tiles
    (24, 512, 7, 7, 512, 1, 4),      # 12 tiles (two filter blocks) x 4 slices of 18
    (10, 256, 28, 28, 256, 2, 2),    # stride 2: phase planes, 8 tiles x 2
    (30, 128, 9, 11, 320, 1, 2),     # ragged filters (rows beyond F in the second filter block), odd rows
])
def test_conv3x3_tap_gemm_split_k(rt, cfg, dt, monkeypatch):
    """The split-K form of the tap GEMM (route "tap_gemm_splitk"): S workgroups per 256 x 256 tile, each over a contiguous range of
    the tile's K-tiles (the cursors start in the middle of a channel block's taps), the slices trading accumulator row blocks
    through the fp32 slab (written through, one flag word per wave pair) and each finishing 8 / S row blocks of every wave.
    IROCM_CONV_TAP_SPLIT forces the factor; run TWICE (the flag words must be zero again after the first launch) and checked
    against the oracle with bias + ReLU."""
    n, c, h, w, f, st, split = cfg
    monkeypatch.setenv("IROCM_CONV_TAP_SPLIT", str(split))
    rng = np.random.default_rng(abs(hash(cfg)) % 2 ** 32)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = (rng.standard_normal((f, c, 3, 3)) / np.sqrt(9 * c)).astype(np.float32)
    b = rng.standard_normal((f,)).astype(np.float32)
    xd, wd, bd = dev_slack2(x, TD[dt]), dev(wt, TD[dt]), dev(b, TD[dt])
    oh, ow = (h + st - 1) // st, (w + st - 1) // st
    outs = []
    try:
        ops.set_conv_variant(rt, 7)
        for rep in range(2):
            guard = torch.full((n, f, oh, ow), 7.0, device="cuda", dtype=TD[dt])
            outs.append(ops.conv2d(rt, xd, wd, 1, 1, st, st, bias=bd, act=1, out=guard))
            assert ops.conv_last_route(rt) == "tap_gemm_splitk"
    finally:
        ops.set_conv_variant(rt, -1)
    want = np.maximum(R.conv2d(R.round_to(x, dt), R.round_to(wt, dt), 1, 1, st, st, 1, 1) + R.round_to(b, dt).reshape(1, f, 1, 1), 0)
    tol = {"f16": 3e-3, "bf16": 2.4e-2}[dt]
    for y in outs:
        got = host(y)
        assert np.isfinite(got).all()
        assert np.allclose(got, want, rtol=tol, atol=tol), np.abs(got - want).max()
    assert torch.equal(outs[0], outs[1])  # fixed summation order: bit-identical run to run


DW_CFGS = [
    # n, c, h, w, mult, k, stride, pad
    (2, 32, 30, 30, 1, 3, 1, 1),     # rows of 30: every row has a ragged last run; whole planes per workgroup
    (3, 24, 19, 19, 1, 5, 2, 2),     # odd planes, stride 2 (output 10 x 10): element stores on odd row starts
    (1, 16, 150, 150, 1, 3, 1, 1),   # EfficientNet-Lite4 stem-side plane: row strips, several strips per plane
    (2, 48, 75, 75, 1, 5, 2, 2),     # 75 -> 38, 5 x 5 / 2
    (2, 12, 17, 23, 2, 3, 1, 1),     # channel multiplier 2 (F = 2 C), non-square
    (1, 8, 9, 40, 1, 5, 1, 2),       # wide and flat
    (4, 20, 7, 7, 1, 3, 1, 1),       # tiny planes: many planes per workgroup
    (2, 10, 16, 16, 1, 3, 2, 1),     # even input, stride 2, pad 1 (the last input row / column is never read by r = 2... it is: 15 = 2*7+1)
    (1, 6, 12, 12, 1, 3, 1, 0),      # no padding (valid convolution)
]


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("mode", ["plain", "bias_relu"])
@pytest.mark.parametrize("cfg", DW_CFGS)
def test_conv_depthwise_kernel(rt, cfg, mode, dt):
    """Round 5: depthwise layers (groups == C; reference: src/operators/conv.cc:47-114 with channel_per_group = 1, cuDNN group
    convolution in src/kernels/cuda/conv.cc:57-168) on their own HBM-bound kernel (csrc/conv_dw.hip, route "depthwise") against the
    oracle and against the generic implicit GEMM (variant 1) they used before. The output buffer is pre-filled (every element must
    be written) and sits inside a larger block whose neighbours must stay untouched."""
    n, c, h, w, mult, k, st, pad = cfg
    f = c * mult
    rng = np.random.default_rng(abs(hash(cfg)) % 2 ** 32)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = (rng.standard_normal((f, 1, k, k)) / k).astype(np.float32)
    b = rng.standard_normal((f,)).astype(np.float32) if "bias" in mode else None
    act = 1 if "relu" in mode else 0
    xd, wd = dev(x, TD[dt]), dev(wt, TD[dt])
    bd = dev(b, TD[dt]) if b is not None else None
    oh, ow = (h + 2 * pad - k) // st + 1, (w + 2 * pad - k) // st + 1
    block = torch.full((n * f * oh * ow + 128,), 7.0, device="cuda", dtype=TD[dt])
    out = block[64: 64 + n * f * oh * ow].view(n, f, oh, ow)
    y = ops.conv2d(rt, xd, wd, pad, pad, st, st, bias=bd, act=act, out=out)
    assert ops.conv_last_route(rt) == "depthwise"
    try:
        ops.set_conv_variant(rt, 1)
        y2 = ops.conv2d(rt, xd, wd, pad, pad, st, st, bias=bd, act=act)
        assert ops.conv_last_route(rt) == "generic"
    finally:
        ops.set_conv_variant(rt, -1)
    want = R.conv2d(R.round_to(x, dt), R.round_to(wt, dt), pad, pad, st, st, 1, 1)
    if b is not None:
        want = want + R.round_to(b, dt).reshape(1, f, 1, 1)
    if act:
        want = np.maximum(want, 0)
    tol = {"f16": 3e-3, "bf16": 2.4e-2}[dt]
    assert np.allclose(host(y), want, rtol=tol, atol=tol), np.abs(host(y) - want).max()
    assert np.allclose(host(y), host(y2), rtol=tol, atol=tol)
    assert torch.all(block[:64] == 7.0).item() and torch.all(block[64 + n * f * oh * ow:] == 7.0).item()


F32_CONV_CFGS = [
    # n, c, h, w, f, r, s, ph, pw, sh, sw, dh, dw
    (2, 16, 14, 14, 48, 3, 3, 1, 1, 1, 1, 1, 1),      # 64^2 tiles, planes of 196 (tiles span images)
    (3, 64, 7, 7, 160, 3, 3, 1, 1, 1, 1, 1, 1),       # 49-pixel planes: element stores, ragged filters
    (2, 32, 28, 28, 64, 3, 3, 1, 1, 2, 2, 1, 1),      # stride 2
    (1, 8, 33, 17, 24, 5, 3, 2, 0, 1, 2, 1, 2),       # asymmetric everything, dilation
    (2, 128, 14, 14, 256, 1, 1, 0, 0, 1, 1, 1, 1),    # unit-stride pointwise (columns across images: 392 = 6 tiles + 8)
    (2, 64, 15, 15, 128, 1, 1, 0, 0, 2, 2, 1, 1),     # strided pointwise: implicit GEMM
    (6, 4, 40, 40, 20, 7, 7, 3, 3, 2, 2, 1, 1),       # 7 x 7 / 2 on 4 channels (K = 196)
    (5, 3, 38, 38, 72, 7, 7, 3, 3, 2, 2, 1, 1),       # the stem's form: 3 channels, K = 147 -> weight rows copied into 148-float rows
    (2, 5, 11, 13, 9, 3, 3, 1, 1, 1, 1, 1, 1),        # K = 45 (padded to 48), 9 filters
    (16, 64, 28, 28, 128, 3, 3, 1, 1, 1, 1, 1, 1),    # enough columns for the 128^2 tiles (98 x 1 tiles ... forced below the CU count: 64^2)
    (64, 64, 28, 28, 128, 3, 3, 1, 1, 1, 1, 1, 1),    # 128^2 tiles
    (4, 48, 12, 12, 80, 3, 3, 1, 1, 1, 1, 1, 1),      # 48 channels: the tap-major image pads every tap to 64 (zero weights, masked loads)
    (2, 100, 9, 9, 40, 3, 2, 1, 0, 1, 1, 2, 1),       # 100 channels (K = 600 in FCRS order; tap-major 128 per tap), dilated rows
]


def test_conv_fp32_pointwise_as_one_gemm_per_image(rt, monkeypatch):
    """The A/B route of unit-stride pointwise fp32 layers (IROCM_CONV32_PW_BATCHED: Y[n] = W . X[n] on the fp32 tile GEMM, zero copy)."""
    monkeypatch.setenv("IROCM_CONV32_PW_BATCHED", "1")
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 64, 14, 14)).astype(np.float32)
    wt = (rng.standard_normal((96, 64, 1, 1)) / 8).astype(np.float32)
    b = rng.standard_normal((96,)).astype(np.float32)
    y = ops.conv2d(rt, dev(x, torch.float32), dev(wt, torch.float32), 0, 0, 1, 1, 1, 1, bias=dev(b, torch.float32), act=1)
    assert ops.conv_last_route(rt) == "batched_gemm32"
    want = np.maximum(R.conv2d(x, wt, 0, 0, 1, 1, 1, 1) + b[None, :, None, None], 0)
    assert np.allclose(host(y), want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("mode", ["plain", "bias_res_relu", "bias_res_relu_t128", "bias_res_relu_split1", "bias_res_relu_split3", "plain_split4"])
@pytest.mark.parametrize("cfg", F32_CONV_CFGS)
def test_conv_fp32_on_the_matrix_cores(rt, cfg, mode, monkeypatch):
    """Round 5: fp32 Conv2d (the dtype of north_star's 1e-4 gate and of the intelcpu baseline; reference: cuDNN implicit GEMM,
    src/kernels/cuda/conv.cc:57-168) as an implicit GEMM on v_mfma_f32_32x32x2_f32 (route "igemm32", pointwise layers included) against the oracle within 1e-4 RELATIVE of the output's scale per element (2e-5 absolute below 1)
    and against the one-output-per-thread kernel it replaces (conv variant 1, route "direct32"). Layers with few tiles split K
    over several workgroups per tile (route "igemm32_splitk": raw sums per slice, one reduce pass with bias / residual / ReLU) — by the
    launcher's rule or forced ("_split<S>"). "_t128" forces the 128 x 128 tile
    form (IROCM_CONV32_TILE; the launcher picks it from eight such tiles per CU on — sizes the dense oracle cannot follow)."""
    n, c, h, w, f, r, s, ph, pw, sh, sw, dh, dw = cfg
    if mode.endswith("_t128"):
        monkeypatch.setenv("IROCM_CONV32_TILE", "2")
    if "_split" in mode:  # split-K forced: K-tile ranges per slice, raw partial sums in the workspace, the reduce pass (1 = never)
        monkeypatch.setenv("IROCM_CONV32_SPLIT", mode[-1])
    rng = np.random.default_rng(abs(hash(cfg)) % 2 ** 32)
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    wt = (rng.standard_normal((f, c, r, s)) / np.sqrt(c * r * s)).astype(np.float32)
    oh, ow = (h - (r - sh) * dh + 2 * ph) // sh, (w - (s - sw) * dw + 2 * pw) // sw
    b = rng.standard_normal((f,)).astype(np.float32) if "bias" in mode else None
    res = rng.standard_normal((n, f, oh, ow)).astype(np.float32) if "res" in mode else None
    act = 1 if "relu" in mode else 0
    xd, wd = dev(x, torch.float32), dev(wt, torch.float32)
    bd = dev(b, torch.float32) if b is not None else None
    rd = dev(res, torch.float32) if res is not None else None
    guard = torch.full((n, f, oh, ow), 7.0, device="cuda", dtype=torch.float32)
    y = ops.conv2d(rt, xd, wd, ph, pw, sh, sw, dh, dw, bias=bd, act=act, residual=rd, out=guard)
    assert ops.conv_last_route(rt) in (("igemm32",) if mode.endswith("split1") else ("igemm32", "igemm32_splitk"))
    try:
        ops.set_conv_variant(rt, 1)
        y1 = ops.conv2d(rt, xd, wd, ph, pw, sh, sw, dh, dw, bias=bd, act=act, residual=rd)
        assert ops.conv_last_route(rt) == "direct32"
    finally:
        ops.set_conv_variant(rt, -1)
    want = R.conv2d(x, wt, ph, pw, sh, sw, dh, dw)
    if b is not None:
        want = want + b.astype(np.float64).reshape(1, f, 1, 1)
    if res is not None:
        want = want + res
    if act:
        want = np.maximum(want, 0)
    got = host(y)
    assert np.all(np.abs(got - want) <= 1e-4 * np.abs(want) + 2e-5), np.abs(got - want).max()
    assert np.allclose(got, host(y1), rtol=1e-4, atol=2e-5)
