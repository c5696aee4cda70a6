"""Data-movement / indexing ops on a real MI355X: bit-exact vs the oracle and the reference KATs."""
import numpy as np
import pytest
import torch
from conftest import kat

from infinitensor_amd import ops
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu
CU = "test/kernels/cuda/"
NPDT = [np.float32, np.float16, np.int64, np.int32, np.int8, np.uint8, np.float64]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


def rnd(shape, dtype, seed=0):
    rng = np.random.default_rng(seed)
    if np.issubdtype(dtype, np.integer):
        return rng.integers(0, 100, shape).astype(dtype)
    return rng.standard_normal(shape).astype(dtype)


def test_transpose_reference_kat(rt):
    y = ops.transpose(rt, dev(R.incremental((1, 2, 3, 4))), (0, 2, 1, 3))
    assert np.array_equal(host(y).ravel(), kat(CU + "test_cuda_transpose.cc", 37, "float"))


@pytest.mark.parametrize("dtype", NPDT)
@pytest.mark.parametrize("shape,perm", [((4, 512, 12, 64), (0, 2, 1, 3)), ((3, 130, 70), (0, 2, 1)), ((65, 33), (1, 0)),
                                        ((2, 3, 4, 5), (3, 2, 1, 0)), ((2, 3, 4, 5, 6), (4, 0, 3, 1, 2)), ((7,), (0,)),
                                        ((2, 1, 3, 1, 4), (3, 4, 1, 0, 2)), ((5, 6, 7), (0, 1, 2)), ((2, 3, 2, 3, 2, 3, 2, 3), (7, 6, 5, 4, 3, 2, 1, 0)),
                                        # batched 2-D transposes whose extents are multiples of the 16-byte vector: the vectorised tile kernel
                                        # (whole tiles, ragged tiles in both directions, K^T of a BERT head, one tile)
                                        ((3, 64, 72), (0, 2, 1)), ((2, 40, 24), (0, 2, 1)), ((24, 512, 64), (0, 2, 1)), ((2, 12, 128, 64), (0, 1, 3, 2)),
                                        ((200, 8), (1, 0)), ((8, 136), (1, 0))])
def test_transpose_bit_exact(rt, shape, perm, dtype):
    x = rnd(shape, dtype)
    assert np.array_equal(host(ops.transpose(rt, dev(x), perm)), R.transpose(x, perm))


@pytest.mark.parametrize("dtype", [np.float16, np.float32, np.int8, np.int64])
@pytest.mark.parametrize("form", ["dense", "y_scalar", "x_scalar", "cond_scalar", "ragged_tail"])
def test_where_flat_forms_bit_exact(rt, dtype, form):
    """Same-shape operands / scalar operands (every operand dense or a one-element broadcast): the vectorised flat kernel,
    incl. a length that is not a multiple of the 16-byte vector — bit-exact against the oracle (where.cu:4-19 semantics)."""
    n = (37, 1000) if form != "ragged_tail" else (3, 1237)
    x, y_ = rnd(n, dtype, 3), rnd(n, dtype, 4)
    cond = np.random.default_rng(5).random(n) > 0.4
    if form == "y_scalar":
        y_ = rnd((1,), dtype, 6)
    elif form == "x_scalar":
        x = rnd((1,), dtype, 7)
    elif form == "cond_scalar":
        cond = np.array([True])
    got = ops.where(rt, dev(x), dev(y_), dev(cond))
    assert np.array_equal(host(got), R.where(x, y_, cond))


def test_gather_reference_kats(rt):
    f = CU + "test_cuda_gather.cc"
    y = ops.gather(rt, dev(kat(f, 184, "float").astype(np.float32).reshape(3, 2)), dev(kat(f, 185, "int").astype(np.int32).reshape(2, 2)), 0)
    assert np.array_equal(host(y).ravel(), kat(f, 200, "float"))
    y = ops.gather(rt, dev(R.incremental((3, 3))), dev(kat(f, 209, "int").astype(np.int32).reshape(1, 2)), 1)
    assert np.array_equal(host(y).ravel(), kat(f, 224, "float"))
    for idt in (np.int32, np.int64):  # :227-274, int32 and int64 indices
        y = ops.gather(rt, dev(R.incremental((2, 4, 2))), dev(np.array([0, 3, 1], dtype=idt).reshape(3, 1)), 1)
        assert np.array_equal(host(y).ravel(), kat(f, 249, "float"))


@pytest.mark.parametrize("dtype", [np.float32, np.float16, np.int8, np.int64])
@pytest.mark.parametrize("idt", [np.int32, np.int64])
def test_gather_embedding_bit_exact(rt, dtype, idt):
    table = rnd((3052, 768), dtype, 1)
    ids = np.random.default_rng(2).integers(-3052, 3052, (4, 128)).astype(idt)  # negative indices wrap
    y = ops.gather(rt, dev(table), dev(ids), 0)
    assert np.array_equal(host(y), R.gather(table, ids, 0))
    x = rnd((3, 7, 5), dtype, 3)
    ix = np.array([[6, 0], [2, -1]], dtype=idt)
    assert np.array_equal(host(ops.gather(rt, dev(x), dev(ix), 1)), R.gather(x, ix, 1))


def test_where_reference_kats(rt):
    f = CU + "test_cuda_where.cc"
    y = ops.where(rt, dev(kat(f, 88, "float").astype(np.float32).reshape(2, 2, 3, 1)), dev(kat(f, 89, "float").astype(np.float32).reshape(2, 2, 3, 1)),
                  dev(kat(f, 90, "uint8_t").astype(np.uint8).reshape(2, 2, 3, 1)))
    assert np.array_equal(host(y).ravel(), kat(f, 91, "float"))
    # :93-101 three-way broadcast
    x = np.array([0, 1, 2, 3, 4, 5], np.float32).reshape(2, 1, 1, 3)
    yv = np.ones((1, 2, 1, 1), np.float32)
    c = np.array([0, 1, 1, 0, 0, 0], np.uint8).reshape(2, 1, 3, 1)
    got = host(ops.where(rt, dev(x), dev(yv), dev(c)))
    assert np.array_equal(got, R.where(x, yv, c))


@pytest.mark.parametrize("dtype", [np.float32, np.float16, np.int64])
def test_where_mask_bit_exact(rt, dtype):
    x, y = rnd((4, 12, 32, 32), dtype, 5), rnd((1,), dtype, 6)
    c = (np.random.default_rng(7).random((4, 1, 1, 32)) > 0.5)
    assert np.array_equal(host(ops.where(rt, dev(x), dev(y), dev(c))), R.where(x, y, c))


@pytest.mark.parametrize("dtype", [np.float32, np.float16, np.float64, np.int32, np.int64])
def test_less_feeds_where(rt, dtype):
    """Less -> Where: the reference's CUDA Less writes bool BYTES into a buffer the graph declares with the operand
    dtype and WhereCuda reads bytes (element_wise.cu:101-131, where.cu:4-19); this backend's comparisons write whole
    elements (as the native-CPU kernels do), so Where reads the condition in ITS dtype (infini_rocm_where_ex, what
    WhereRocm passes) and the chain gives the same answer; -0.0 counts as false."""
    a, b = rnd((4, 1, 6), dtype, 1), rnd((1, 5, 6), dtype, 2)
    x, y = rnd((4, 5, 6), dtype, 3), rnd((6,), dtype, 4)
    c = ops.binary(rt, "less", dev(a), dev(b))
    assert c.dtype == dev(a).dtype
    got = host(ops.where(rt, dev(x), dev(y), c))
    assert np.array_equal(got, np.where(a < b, x, np.broadcast_to(y, (4, 5, 6))))
    if np.issubdtype(dtype, np.floating):
        cz = np.array([0.0, -0.0, 1.0, -1.0, np.nan, 1e-30], dtype)
        got = host(ops.where(rt, dev(np.ones(6, dtype)), dev(np.zeros(6, dtype)), dev(cz)))
        assert np.array_equal(got, np.array([0, 0, 1, 1, 1, 1 if dtype != np.float16 else 0], dtype))


def test_concat_split_slice_pad_expand_reference_kats(rt):
    t1 = R.incremental((2, 2, 3, 1))
    y = ops.concat(rt, [dev(t1), dev(R.ones((2, 2, 1, 1))), dev(R.ones((2, 2, 2, 1)))], 2)
    assert np.array_equal(host(y).ravel(), kat(CU + "test_cuda_concat.cc", 93, "float"))
    outs = ops.split(rt, dev(R.incremental((2, 10, 2, 1))), 1, [3, 3, 4])
    for o, line in zip(outs, (35, 37, 38)):
        assert np.array_equal(host(o).ravel(), kat(CU + "test_cuda_split.cc", line, "float"))
    y = ops.slice_(rt, dev(R.incremental((3, 2, 1, 5))), [1, 1], [2, 5], [0, 3])
    assert np.array_equal(host(y).ravel(), kat(CU + "test_cuda_slice.cc", 38, "float"))
    y = ops.pad(rt, dev(R.incremental((1, 2, 3, 2))), [1, 0, 0, 0, 1, 0, 0, 1])
    assert np.array_equal(host(y).ravel(), kat(CU + "test_cuda_pad.cc", 36, "float"))
    y = ops.expand(rt, dev(R.incremental((2, 1, 2, 1))), (2, 2, 2, 3))
    assert np.array_equal(host(y).ravel(), kat(CU + "test_cuda_expand.cc", 37, "float"))


@pytest.mark.parametrize("dtype", [np.float32, np.float16, np.int8, np.int64])
def test_concat_split_roundtrip(rt, dtype):
    """split(concat(xs)) == xs, any axis (size-independent property), ragged sizes incl. an empty part."""
    xs = [rnd((3, n, 5, 2), dtype, n) for n in (4, 1, 0, 7)]
    cat = ops.concat(rt, [dev(x) for x in xs], 1)
    assert np.array_equal(host(cat), R.concat(xs, 1))
    back = ops.split(rt, cat, 1, [4, 1, 0, 7])
    for b, x in zip(back, xs):
        assert np.array_equal(host(b), x)
    xs = [rnd((2, 3, n), dtype, n) for n in (3, 5)]
    assert np.array_equal(host(ops.concat(rt, [dev(x) for x in xs], -1)), R.concat(xs, 2))


@pytest.mark.parametrize("dtype,rows", [(np.float16, 70003), (np.int8, 131075), (np.float32, 66000)])
def test_concat_split_tall_tensors_take_four_rows_per_trip(rt, dtype, rows):
    """Round 5: tensors with many more rows than the copy grid take the four-consecutive-rows-per-trip form of the Concat / Split
    kernel (all loads of a trip before its stores); row counts that are not multiples of four end in a partial group; three inputs of
    different widths; bit-exact both ways."""
    xs = [rnd((rows, n), dtype, n) for n in (8, 24, 16)]
    cat = ops.concat(rt, [dev(x) for x in xs], 1)
    want = R.concat(xs, 1)
    assert np.array_equal(host(cat), want)
    back = ops.split(rt, cat, 1, [8, 24, 16])
    for b, x in zip(back, xs):
        assert np.array_equal(host(b), x)


@pytest.mark.parametrize("dtype", [np.float32, np.float16, np.int64])
def test_slice_pad_bit_exact(rt, dtype):
    x = rnd((4, 6, 7, 9), dtype, 11)
    y = ops.slice_(rt, dev(x), [1, 2, -5], [3, 7, 100], [0, 2, 3], [1, 2, 3])
    assert np.array_equal(host(y), R.slice_(x, [1, 2, -5], [3, 7, 100], [0, 2, 3], [1, 2, 3]))
    p = [0, 1, 2, 0, 3, 0, 1, 4]
    assert np.array_equal(host(ops.pad(rt, dev(x), p)), R.pad(x, p))
    # pad then slice back is the identity
    z = ops.slice_(rt, ops.pad(rt, dev(x), p), [0, 1, 2, 0], [4, 7, 9, 9])
    assert np.array_equal(host(z), x)


@pytest.mark.parametrize("dtype", [np.float32, np.float16, np.int32])
def test_expand_reshape_bit_exact(rt, dtype):
    x = rnd((3, 1, 5), dtype, 13)
    assert np.array_equal(host(ops.expand(rt, dev(x), (2, 3, 4, 5))), R.expand(x, (2, 3, 4, 5)))
    assert np.array_equal(host(ops.reshape(rt, dev(x), (5, 3))), x.reshape(5, 3))


def test_gather_elements_kats_and_random(rt):
    """test_cuda_gather_elements.cc:10-42 + random cases vs the oracle (int32 / int64 indices, negative indices)."""
    GE = "test/kernels/cuda/test_cuda_gather_elements.cc"
    d = torch.from_numpy(kat(GE, 19).astype(np.int32).reshape(3, 3)).cuda()
    i = torch.from_numpy(kat(GE, 20).astype(np.int64).reshape(2, 3)).cuda()
    assert ops.gather_elements(rt, d, i, 0).cpu().numpy().ravel().tolist() == [4, 8, 3, 7, 2, 3]
    d = torch.from_numpy(kat(GE, 36).astype(np.float32).reshape(2, 2)).cuda()
    i = torch.from_numpy(kat(GE, 37).astype(np.int32).reshape(2, 2)).cuda()
    assert ops.gather_elements(rt, d, i, 1).cpu().numpy().ravel().tolist() == [1., 1., 4., 3.]
    rng = np.random.default_rng(2)
    for shape, ishape, axis, idt in (((5, 7, 9), (5, 4, 9), 1, np.int64), ((6, 33), (6, 33), 1, np.int32),
                                     ((4, 3, 2, 8), (2, 3, 2, 8), 0, np.int64), ((17,), (40,), 0, np.int32)):
        data = rng.standard_normal(shape).astype(np.float16)
        idx = rng.integers(-shape[axis], shape[axis], ishape).astype(idt)
        got = ops.gather_elements(rt, torch.from_numpy(data).cuda(), torch.from_numpy(idx).cuda(), axis).cpu().numpy()
        assert np.array_equal(got, R.gather_elements(data, idx, axis))


def test_extend_kat_and_depth_to_space(rt):
    """test_cuda_extend.cc:12-43; DepthToSpace (no reference test: ONNX definition in numpy)."""
    x = torch.arange(24, dtype=torch.float32).reshape(2, 3, 2, 2).cuda()
    y = ops.extend(rt, x, 1, 1)
    assert tuple(y.shape) == (2, 6, 2, 2)
    assert R.equal_data(y.cpu().numpy().ravel(), kat("test/kernels/cuda/test_cuda_extend.cc", 37, "float"))
    rng = np.random.default_rng(3)
    a = rng.standard_normal((2, 12, 5, 7)).astype(np.float16)
    for mode in ("DCR", "CRD"):
        got = ops.depth_to_space(rt, torch.from_numpy(a).cuda(), 2, mode).cpu().numpy()
        assert np.array_equal(got, R.depth_to_space(a, 2, mode))


def test_resize_reference_kats_and_random(rt):
    """test_cuda_resize.cc (21 stretch-policy cases, tests/resize_cases.py) through the C ABI, then random shapes vs
    the oracle in f32 / f16 for every mode x coordinate transform."""
    from resize_cases import CASES, materialise

    for case in CASES:
        x, out, scales, roi, want = materialise(case)
        y = ops.resize(rt, torch.from_numpy(x).cuda(), out, scales, case[4], case[5], case[6], roi)
        assert R.equal_data(y.cpu().numpy().ravel(), want, 1e-5), case
    rng = np.random.default_rng(6)
    x = rng.standard_normal((2, 3, 7, 9)).astype(np.float32)
    for mode in ("nearest", "linear", "cubic"):
        for coord in ("half_pixel", "pytorch_half_pixel", "align_corners", "asymmetric"):
            for out in ((2, 3, 13, 5), (2, 3, 4, 20)):
                scales = [o / i for o, i in zip(out, x.shape)]
                want = R.resize(x, out, scales, mode, coord, "floor")
                got = ops.resize(rt, torch.from_numpy(x).cuda(), out, scales, mode, coord, "floor").cpu().numpy()
                assert np.allclose(got, want, rtol=1e-4, atol=1e-5), (mode, coord, out)
                got16 = ops.resize(rt, torch.from_numpy(x).half().cuda(), out, scales, mode, coord, "floor").float().cpu().numpy()
                assert np.allclose(got16, R.resize(x.astype(np.float16), out, scales, mode, coord, "floor"), rtol=4e-3, atol=4e-3)
