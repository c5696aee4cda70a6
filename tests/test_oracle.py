"""Pin the CPU oracle (oracle/ref_ops.py): it must reproduce
  (1) the golden vectors asserted by the reference's own kernel tests (tests/golden/kats.json), with
      the reference's comparator (equalData: 1e-6 relative, include/core/tensor.h:197-234), and
  (2) the reference native-CPU backend built from /root/reference (oracle/_ref) where that backend
      implements the op.
CPU only — runs in the build container and on the GPU box alike.
"""
import numpy as np
import pytest
from conftest import kat

from oracle import ref_ops as R

CU = "test/kernels/cuda/"
MKL = "test/kernels/intelcpu/"
NCPU = "test/kernels/nativecpu/"


def eq(a, b, rel=1e-6):
    return R.equal_data(np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel(), rel)


# ---- MatMul: test_cuda_matmul.cc:47-66 (== test_mkl_matmul.cc:32-40) --------------------------------
MATMUL_CASES = [
    # (genA, genB, transA, transB, shapeA, shapeB, golden line)
    ("inc", "one", False, False, (1, 3, 5), (1, 5, 2), 50),
    ("inc", "inc", True, False, (2, 3, 4), (2, 3, 2), 53),
    ("inc", "inc", False, False, (2, 3, 5), (5, 2), 58),
    ("inc", "inc", True, False, (2, 5, 3), (5, 2), 61),
    ("inc", "inc", False, False, (3, 5), (5, 2), 65),
]


def gen(kind, shape, dtype=np.float32):
    return {"inc": R.incremental, "one": R.ones}[kind](shape, dtype)


@pytest.mark.parametrize("case", MATMUL_CASES)
def test_matmul_kats(case):
    ga, gb, ta, tb, sa, sb, line = case
    want = kat(CU + "test_cuda_matmul.cc", line, "float")
    got = R.matmul(gen(ga, sa), gen(gb, sb), None, ta, tb)
    assert eq(got, want)


def test_matmul_kats_mkl_file_identical():
    assert np.array_equal(kat(MKL + "test_mkl_matmul.cc", 35, "float"), kat(CU + "test_cuda_matmul.cc", 50, "float"))
    assert np.array_equal(kat(MKL + "test_mkl_matmul.cc", 38, "float"), kat(CU + "test_cuda_matmul.cc", 53, "float"))


def test_matmul_vs_reference_native_cpu(ref_backend):
    """Reference NaiveMatmul (src/kernels/cpu/matmul.cc:6-25; no batch/transpose support)."""
    rng = np.random.default_rng(0)
    a = rng.uniform(-1, 1, (1, 64, 48)).astype(np.float32)
    b = rng.uniform(-1, 1, (1, 48, 40)).astype(np.float32)
    h = ref_backend.GraphHandler(ref_backend.cpu_runtime())
    ta, tb = h.tensor(list(a.shape), 1), h.tensor(list(b.shape), 1)
    tc = h.matmul(ta, tb, None, False, False, None, ref_backend.ActType.Linear, "default")
    h.data_malloc()
    ta.copyin_numpy(a)
    tb.copyin_numpy(b)
    h.run()
    got = np.array(tc.copyout_float()).reshape(1, 64, 40)
    assert np.allclose(got, R.matmul(a, b), rtol=1e-5, atol=1e-5)


# ---- Softmax: test_cuda_softmax.cc:67-132 ------------------------------------------------------------
SM = CU + "test_cuda_softmax.cc"


@pytest.mark.parametrize("axis,in_line,out_line", [(0, 70, 73), (1, 83, 86), (2, 96, 99), (3, 107, 110)])
def test_softmax_kats_fp32(axis, in_line, out_line):
    x = kat(SM, in_line, "float").astype(np.float32).reshape(2, 3, 2, 2)
    assert np.array_equal(x.ravel(), np.arange(24))
    assert eq(R.softmax(x, axis), kat(SM, out_line, "float"))


def test_softmax_kats_fp16():
    # test_cuda_softmax.cc:117-131: ValGenerator<2> input, axes 0 and 1, fp16 storage
    x = R.value((2, 3, 2, 2), 2.0, np.float16)
    assert eq(R.softmax(x, 0).astype(np.float16), kat(SM, 119, "float"), 1e-3)
    assert eq(R.softmax(x, 1).astype(np.float16), kat(SM, 126, "float"), 1e-3)  # "data accuracy down"


# ---- LayerNorm: test_cuda_layernorm.cc:150-222 -------------------------------------------------------
LN = CU + "test_cuda_layernorm.cc"
# (input line, scale line, expected line, bias line or None)
LN_CASES = [(153, 157, 158, 165), (168, 172, 173, 180), (183, 187, 188, 195), (198, 202, 203, None)]


@pytest.mark.parametrize("xi,si,yi,bi", LN_CASES)
def test_layernorm_kats(xi, si, yi, bi):
    x = kat(LN, xi, "float").astype(np.float32).reshape(2, 3, 2, 3)
    scale = kat(LN, si, "float")
    bias = kat(LN, bi, "float") if bi else None
    assert eq(R.layer_norm(x, scale, bias, 1e-5, 3), kat(LN, yi, "float"))


def test_layernorm_kat_fp16():
    # :213-220 ValGenerator<2> for input, scale and bias -> 2 everywhere
    y = R.layer_norm(R.value((2, 3, 2, 3), 2, np.float16), R.value((3,), 2), R.value((3,), 2), 1e-5, 3)
    assert eq(y, kat(LN, 216, "float"))


# ---- Element-wise: test_cuda_element_wise.cc:47-69, test_nativecpu_elementwise.cc:29-42 --------------
EW_CASES = [("add", "inc", (1, 2, 2, 3), 50), ("sub", "inc", (1, 2, 2, 3), 53), ("mul", "inc", (1, 2, 2, 3), 56),
            ("div", "one", (1, 2, 2, 3), 60), ("min", "inc", (1, 2, 2, 3), 63), ("max", "inc", (1, 2, 2, 3), 66),
            ("pow", "inc", (1, 2, 2, 1), 68)]


@pytest.mark.parametrize("op,g,shape,line", EW_CASES)
def test_elementwise_kats(op, g, shape, line):
    a = gen(g, shape)
    assert eq(R.binary(op, a, a), kat(CU + "test_cuda_element_wise.cc", line, "float"))


@pytest.mark.parametrize("op,gb,line", [("add", "inc", 32), ("mul", "inc", 35), ("sub", "inc", 38), ("div", "one", 41)])
def test_elementwise_rank5_broadcast_kats(op, gb, line):
    a, b = gen("inc", (1, 2, 2, 3, 1)), gen(gb, (2, 1, 1))
    assert eq(R.binary(op, a, b), kat(NCPU + "test_nativecpu_elementwise.cc", line, "float"))


def test_elementwise_vs_reference_native_cpu(ref_backend):
    rng = np.random.default_rng(1)
    a = rng.uniform(0.5, 2, (2, 3, 4, 5)).astype(np.float32)
    b = rng.uniform(0.5, 2, (3, 1, 5)).astype(np.float32)
    for name in ("add", "sub", "mul", "div"):  # Pow has no native-CPU kernel
        h = ref_backend.GraphHandler(ref_backend.cpu_runtime())
        ta, tb = h.tensor(list(a.shape), 1), h.tensor(list(b.shape), 1)
        tc = getattr(h, name)(ta, tb, None)
        h.data_malloc()
        ta.copyin_numpy(a)
        tb.copyin_numpy(b)
        h.run()
        got = np.array(tc.copyout_float()).reshape(2, 3, 4, 5)
        assert np.allclose(got, R.binary(name, a, b), rtol=2e-6, atol=1e-6), name


# ---- Unary: differential vs the reference native-CPU kernel (test_cuda_unary.cc:12-41,122-143) ------
UNARY_REF = ["relu", "silu", "abs", "sigmoid", "tanh", "hardSigmoid", "hardSwish", "sqrt", "neg", "erf", "gelu"]
_PY = {"hardSigmoid": "hard_sigmoid", "hardSwish": "hard_swish"}


@pytest.mark.parametrize("name", UNARY_REF)
def test_unary_vs_reference_native_cpu(ref_backend, name):
    x = R.incremental((1, 2, 2, 3))
    h = ref_backend.GraphHandler(ref_backend.cpu_runtime())
    tx = h.tensor([1, 2, 2, 3], 1)
    ty = getattr(h, name)(tx, None)
    h.data_malloc()
    tx.copyin_numpy(x)
    h.run()
    got = np.array(ty.copyout_float())
    assert eq(got, R.unary(_PY.get(name, name), x), 2e-6), name


def test_unary_kats():
    f = CU + "test_cuda_unary.cc"
    x = kat(f, 77, "float")
    assert eq(R.unary("leaky_relu", x, 0.01), kat(f, 95, "float"))          # LeakyRelu alpha 0.01 (:71-97)
    assert eq(R.unary("elu", R.incremental((2, 2, 3, 1)), 1.0), kat(f, 119, "float"))  # Elu (:99-120)


def test_cast_semantics():
    x = np.array([-1.7, -0.2, 0.0, 0.9, 2.5, 300.0], dtype=np.float32)
    assert np.array_equal(R.cast(x, np.int32), np.array([-1, 0, 0, 0, 2, 300], dtype=np.int32))
    assert np.array_equal(R.cast(np.arange(8, dtype=np.float32), np.float16), np.arange(8, dtype=np.float16))


# ---- bf16 rounding helper agrees with torch's RNE --------------------------------------------------
def test_bf16_rounding_matches_torch():
    import torch

    rng = np.random.default_rng(3)
    x = rng.normal(size=4096).astype(np.float32)
    want = torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(R.f32_to_bf16_bits(x), want)
