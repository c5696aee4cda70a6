"""CPU ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.

NumPy (fp64 accumulate) restatement of the reference's operator semantics for the hot path
`RuntimeObj::run -> Kernel::compute`. Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module; infinitensor_amd/ never does.

Pinning: every function here is checked in tests/test_oracle.py against
  (1) the golden vectors of the reference's own tests (tests/golden/kats.json, extracted by
      tests/golden/extract_kats.py from /root/reference/test/kernels/{cuda,intelcpu,nativecpu}), and
  (2) the reference's native-CPU backend compiled from /root/reference (oracle/_ref, built by
      oracle/build_ref.py) wherever that backend implements the op correctly.
The intelcpu backend named in north_star cannot be built here (needs dpcpp/oneDNN v2/oneMKL);
its tests share their golden vectors verbatim with the CUDA tests used above.

Each function cites the reference file:line whose semantics it restates. Float results are
computed in float64 and returned in the input dtype unless stated.
"""
from __future__ import annotations

import math
from typing import Sequence

import numpy as np

try:  # ml_dtypes is not installed in this image; bf16 goes through uint16 bit patterns.
    import ml_dtypes  # type: ignore  # noqa: F401
except Exception:  # pragma: no cover
    ml_dtypes = None


# ------------------------------------------------------------------------------------------------
# data generators (reference: include/utils/data_generator.h:30-102)
# ------------------------------------------------------------------------------------------------
def incremental(shape, dtype=np.float32):
    """IncrementalGenerator: 0, 1, 2, ..."""
    return np.arange(int(np.prod(shape)), dtype=np.float64).astype(dtype).reshape(shape)


def ones(shape, dtype=np.float32):
    """OneGenerator."""
    return np.ones(shape, dtype=dtype)


def value(shape, v, dtype=np.float32):
    """ValGenerator<v>."""
    return np.full(shape, v, dtype=dtype)


# ------------------------------------------------------------------------------------------------
# bf16 helpers (round-to-nearest-even; reference has no bf16 kernel: SURVEY fact 4)
# ------------------------------------------------------------------------------------------------
def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    r[nan] = ((u[nan] >> 16) | 0x40).astype(np.uint16)
    return r


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


def round_to(x: np.ndarray, dtype: str) -> np.ndarray:
    """Round an fp64/fp32 array to the storage precision named by dtype, return float64 values."""
    if dtype in ("f32", "float32"):
        return x.astype(np.float32).astype(np.float64)
    if dtype in ("f16", "float16"):
        return x.astype(np.float16).astype(np.float64)
    if dtype in ("bf16", "bfloat16"):
        return bf16_bits_to_f32(f32_to_bf16_bits(x.astype(np.float32))).astype(np.float64)
    raise ValueError(dtype)


# ------------------------------------------------------------------------------------------------
# comparator (reference: TensorObj::equalData, include/core/tensor.h:197-234)
# ------------------------------------------------------------------------------------------------
def equal_data(a: np.ndarray, b: np.ndarray, rel: float = 1e-6) -> bool:
    """Reference comparator: relative error <= rel (absolute when one side is 0); exact for ints."""
    a = np.asarray(a)
    b = np.asarray(b)
    if a.shape != b.shape and a.size == b.size:
        b = b.reshape(a.shape)
    if np.issubdtype(a.dtype, np.integer) or a.dtype == np.bool_:
        return bool(np.array_equal(a, b))
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    zero = (np.minimum(np.abs(a), np.abs(b)) == 0)
    err_abs = np.abs(a - b)
    denom = np.maximum(np.abs(a), np.abs(b))
    with np.errstate(divide="ignore", invalid="ignore"):
        err_rel = np.where(denom > 0, err_abs / denom, 0.0)
    bad = np.where(zero, err_abs > rel, err_rel > rel)
    return not bool(bad.any())


# ------------------------------------------------------------------------------------------------
# MatMul (reference: src/operators/matmul.cc:26-49 shape rule; src/kernels/cuda/matmul.cc:67-174
# batch broadcast + bias expanded into C with beta = 1)
# ------------------------------------------------------------------------------------------------
def matmul(a: np.ndarray, b: np.ndarray, bias: np.ndarray | None = None, trans_a: bool = False,
           trans_b: bool = False) -> np.ndarray:
    A = np.asarray(a, dtype=np.float64)
    B = np.asarray(b, dtype=np.float64)
    if trans_a:
        A = np.swapaxes(A, -1, -2)
    if trans_b:
        B = np.swapaxes(B, -1, -2)
    assert A.shape[-1] == B.shape[-2], "reference: IT_ASSERT(kA == kB)"
    C = np.matmul(A, B)  # numpy applies the same leading-dim broadcast as infer_broadcast
    if bias is not None:
        C = C + np.asarray(bias, dtype=np.float64)
    return C


# ------------------------------------------------------------------------------------------------
# Softmax (ONNX Softmax-13 along `axis`; reference kernel: src/kernels/cuda/softmax.cu:8-17 online
# max/sum merge, glue softmax.cc:9-31). The native-CPU kernel ignores axis (cpu/unary.cc:178-194)
# and is NOT used as an oracle for this op.
# ------------------------------------------------------------------------------------------------
def softmax(x: np.ndarray, axis: int) -> np.ndarray:
    X = np.asarray(x, dtype=np.float64)
    m = X.max(axis=axis, keepdims=True)
    e = np.exp(X - m)
    return e / e.sum(axis=axis, keepdims=True)


# ------------------------------------------------------------------------------------------------
# LayerNormalization (ONNX-17: normalise over axis..rank-1; reference kernel
# src/kernels/cuda/layer_norm.cu:4-148 normalises dims[axis] only — identical when axis is the
# last dim, which is what every reference test uses: test_cuda_layernorm.cc:150-222).
# scale/bias: full-length or scalar broadcast (layer_norm.cu:41-89).
# ------------------------------------------------------------------------------------------------
def layer_norm(x: np.ndarray, scale: np.ndarray, bias: np.ndarray | None, eps: float, axis: int = -1) -> np.ndarray:
    X = np.asarray(x, dtype=np.float64)
    axis = axis % X.ndim
    red = tuple(range(axis, X.ndim))
    mu = X.mean(axis=red, keepdims=True)
    var = ((X - mu) ** 2).mean(axis=red, keepdims=True)
    y = (X - mu) / np.sqrt(var + eps)
    s = np.asarray(scale, dtype=np.float64)
    y = y * (s.reshape(X.shape[axis:]) if s.size > 1 else s.reshape(()))
    if bias is not None:
        b = np.asarray(bias, dtype=np.float64)
        y = y + (b.reshape(X.shape[axis:]) if b.size > 1 else b.reshape(()))
    return y


# RMSNorm (reference: src/kernels/cuda/rms_norm.cu:35-54; eps hard-coded 1e-5 at :46)
def rms_norm(x: np.ndarray, w: np.ndarray, eps: float = 1e-5) -> np.ndarray:
    X = np.asarray(x, dtype=np.float64)
    ms = (X * X).mean(axis=-1, keepdims=True)
    return X / np.sqrt(ms + eps) * np.asarray(w, dtype=np.float64)


# ------------------------------------------------------------------------------------------------
# Binary element-wise with numpy broadcast (reference: src/kernels/cpu/element_wise.cc:43-112;
# broadcast rule src/utils/operator_utils.cc:6-32). Comparison ops return 1/0 in the input dtype
# (`(T)(val0 < val1)`, element_wise.cc:30-41).
# ------------------------------------------------------------------------------------------------
def binary(op: str, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.asarray(a)
    b = np.asarray(b)
    isint = np.issubdtype(a.dtype, np.integer)
    A = a if isint else a.astype(np.float64)
    B = b if isint else b.astype(np.float64)
    if op == "add": r = A + B
    elif op == "sub": r = A - B
    elif op == "mul": r = A * B
    elif op == "div":
        if isint:
            # C integer division truncates toward zero (reference: `(T)(val0 / val1)`)
            q = np.abs(A) // np.maximum(np.abs(B), 1)
            r = np.where(B == 0, 0, np.sign(A) * np.sign(B) * q).astype(a.dtype)
        else:
            with np.errstate(divide="ignore", invalid="ignore"):
                r = A / B
    elif op == "pow":
        r = np.power(A, B) if not isint else np.power(A.astype(np.int64), B.astype(np.int64)).astype(a.dtype)
    elif op == "min": r = np.minimum(A, B)
    elif op == "max": r = np.maximum(A, B)
    elif op == "equal": r = (A == B)
    elif op == "greater": r = (A > B)
    elif op == "greater_equal": r = (A >= B)
    elif op == "less": r = (A < B)
    elif op == "less_equal": r = (A <= B)
    else:
        raise ValueError(op)
    if r.dtype == np.bool_:
        r = r.astype(a.dtype if isint else np.float64)
    return r


# ------------------------------------------------------------------------------------------------
# Unary (formulas: src/kernels/cpu/unary.cc:8-72; Gelu = 0.5 x (1 + erf(x / sqrt 2)) :44-46;
# HardSigmoid :15-17; HardSwish :19-22; Silu :48-50; CUDA side src/kernels/cuda/unary.cu:31-143)
# ------------------------------------------------------------------------------------------------
_erf = np.vectorize(math.erf, otypes=[np.float64])


def unary(op: str, x: np.ndarray, p0: float = float("nan"), p1: float = float("nan")) -> np.ndarray:
    X = np.asarray(x, dtype=np.float64)
    with np.errstate(all="ignore"):
        if op == "relu": return np.maximum(X, 0.0)
        if op == "sigmoid": return 1.0 / (1.0 + np.exp(-X))
        if op == "tanh": return np.tanh(X)
        if op == "abs": return np.abs(X)
        if op == "sqrt": return np.sqrt(X)
        if op == "gelu": return 0.5 * X * (1.0 + _erf(X / math.sqrt(2.0)))
        if op == "silu": return X / (1.0 + np.exp(-X))
        if op == "neg": return -X
        if op == "erf": return _erf(X)
        if op == "hard_sigmoid": return np.maximum(0.0, np.minimum(1.0, 0.2 * X + 0.5))
        if op == "hard_swish": return X * np.maximum(0.0, np.minimum(1.0, X / 6.0 + 0.5))
        if op == "exp": return np.exp(X)
        if op == "log": return np.log(X)
        if op == "reciprocal": return 1.0 / X
        if op == "elu": return np.where(X >= 0, X, p0 * (np.exp(X) - 1.0))
        if op == "leaky_relu": return np.where(X >= 0, X, p0 * X)
        if op == "clip":
            r = X
            if not math.isnan(p0): r = np.maximum(r, p0)
            if not math.isnan(p1): r = np.minimum(r, p1)
            return r
        if op == "sin": return np.sin(X)
        if op == "cos": return np.cos(X)
        if op == "ceil": return np.ceil(X)
        if op == "floor": return np.floor(X)
        if op == "round": return np.rint(X)
    raise ValueError(op)


# Cast (reference: src/kernels/cuda/unary.cc:30-68; C cast semantics: float->int truncates)
def cast(x: np.ndarray, dst) -> np.ndarray:
    x = np.asarray(x)
    dst = np.dtype(dst)
    if np.issubdtype(dst, np.integer) and np.issubdtype(x.dtype, np.floating):
        return np.trunc(x).astype(dst)
    if dst == np.bool_:
        return x != 0
    return x.astype(dst)


# ------------------------------------------------------------------------------------------------
# Conv2d (reference: src/operators/conv.cc:47-114 shape rule, index math src/kernels/cpu/conv.cc:25-50:
# cross-correlation NCHW x FCRS, symmetric zero pad, stride, dilation, groups)
# ------------------------------------------------------------------------------------------------
def conv2d(x: np.ndarray, w: np.ndarray, ph: int, pw: int, sh: int, sw: int, dh: int, dw: int) -> np.ndarray:
    X = np.asarray(x, dtype=np.float64)
    W = np.asarray(w, dtype=np.float64)
    n, c, h, wd = X.shape
    f, cpg, r, s = W.shape
    g = c // cpg
    fpg = f // g
    oh = (h - (r - sh) * dh + ph * 2) // sh  # reference conv.cc:98-101
    ow = (wd - (s - sw) * dw + pw * 2) // sw
    Xp = np.zeros((n, c, h + 2 * ph, wd + 2 * pw), dtype=np.float64)
    Xp[:, :, ph:ph + h, pw:pw + wd] = X
    Y = np.zeros((n, f, oh, ow), dtype=np.float64)
    for gi in range(g):
        xs = Xp[:, gi * cpg:(gi + 1) * cpg]
        ws = W[gi * fpg:(gi + 1) * fpg]
        for ir in range(r):
            for is_ in range(s):
                patch = xs[:, :, ir * dh: ir * dh + (oh - 1) * sh + 1: sh, is_ * dw: is_ * dw + (ow - 1) * sw + 1: sw]
                Y[:, gi * fpg:(gi + 1) * fpg] += np.einsum("nchw,fc->nfhw", patch, ws[:, :, ir, is_], optimize=True)
    return Y


def conv2d_at(x: np.ndarray, w: np.ndarray, coords: np.ndarray, ph: int, pw: int, sh: int, sw: int, dh: int, dw: int) -> np.ndarray:
    """conv2d above at sampled output positions only: coords [k, 4] = (n, f, oh, ow) -> [k] values, one C/g * R * S dot
    product each in fp64 — the checker for full-size layers (batch 128) whose whole output the dense form cannot produce in
    seconds. Index math as the reference's naive kernel (src/kernels/cpu/conv.cc:25-50: posH = h * sh - ph + r * dh,
    out-of-range taps contribute zero; filter f reads channel group f / (F / g))."""
    X = np.asarray(x)
    W = np.asarray(w, dtype=np.float64)
    n, c, h, wd = X.shape
    f, cpg, r, s = W.shape
    g = c // cpg
    fpg = f // g
    co = np.asarray(coords, dtype=np.int64)
    ni, fi, oy, ox = co[:, 0], co[:, 1], co[:, 2], co[:, 3]
    k = co.shape[0]
    acc = np.zeros(k, dtype=np.float64)
    cidx = (fi // fpg)[:, None] * cpg + np.arange(cpg)[None, :]  # [k, cpg]
    for ir in range(r):
        iy = oy * sh - ph + ir * dh
        for is_ in range(s):
            ix = ox * sw - pw + is_ * dw
            ok = (iy >= 0) & (iy < h) & (ix >= 0) & (ix < wd)
            if not ok.any():
                continue
            iyc, ixc = np.clip(iy, 0, h - 1), np.clip(ix, 0, wd - 1)
            xv = X[ni[:, None], cidx, iyc[:, None], ixc[:, None]].astype(np.float64)  # [k, cpg]
            acc += np.where(ok, np.einsum("kc,kc->k", xv, W[fi, :, ir, is_]), 0.0)
    return acc


# ------------------------------------------------------------------------------------------------
# Pooling (reference: src/operators/pooling.cc:17-35 output size with ceil_mode; kernels
# src/kernels/cuda/pooling.cc:6-95: MaxPool pads with -inf, AveragePool = COUNT_INCLUDE_PADDING
# (:86-90); dilation is accepted but ignored by cuDNN path — we honour it for MaxPool like the
# native CPU kernel src/kernels/cpu/pooling.cc)
# ------------------------------------------------------------------------------------------------
def pool2d(x: np.ndarray, kind: str, kh: int, kw: int, dh: int, dw: int, ph: int, pw: int, sh: int, sw: int,
           ceil_mode: int = 0) -> np.ndarray:
    X = np.asarray(x, dtype=np.float64)
    n, c, h, w = X.shape

    def osz(i, k, d, p, s):
        num = i + 2 * p - d * (k - 1) - 1
        return (int(math.ceil(num / s)) if ceil_mode else num // s) + 1

    oh, ow = osz(h, kh, dh, ph, sh), osz(w, kw, dw, pw, sw)
    Y = np.zeros((n, c, oh, ow), dtype=np.float64)
    for i in range(oh):
        for j in range(ow):
            vals = []
            for r in range(kh):
                for s in range(kw):
                    y_, x_ = i * sh - ph + r * dh, j * sw - pw + s * dw
                    if 0 <= y_ < h and 0 <= x_ < w:
                        vals.append(X[:, :, y_, x_])
                    elif kind == "avg":
                        vals.append(np.zeros((n, c)))
            if kind == "max":
                Y[:, :, i, j] = np.max(np.stack(vals, 0), axis=0) if vals else -np.inf
            else:
                Y[:, :, i, j] = np.sum(np.stack(vals, 0), axis=0) / (kh * kw)
    return Y


# BatchNormalization inference (reference: src/kernels/cuda/batch_norm.cc:7-67, per channel dim 1;
# operator input order X, mean, var, scale, bias — include/operators/batch_norm.h:10-50)
def batch_norm(x, mean, var, scale, bias, eps: float) -> np.ndarray:
    X = np.asarray(x, dtype=np.float64)
    shp = [1, -1] + [1] * (X.ndim - 2)
    m, v, s, b = (np.asarray(t, dtype=np.float64).reshape(shp) for t in (mean, var, scale, bias))
    return (X - m) / np.sqrt(v + eps) * s + b


# ReduceMean / ReduceSum (reference: src/kernels/cuda/reduce.cc:10-108; src/operators/reduce.cc)
def reduce(kind: str, x: np.ndarray, axes: Sequence[int], keepdims: bool) -> np.ndarray:
    X = np.asarray(x, dtype=np.float64)
    ax = tuple(a % X.ndim for a in axes) if len(axes) else tuple(range(X.ndim))
    return X.mean(axis=ax, keepdims=keepdims) if kind == "mean" else X.sum(axis=ax, keepdims=keepdims)


# ------------------------------------------------------------------------------------------------
# Data movement (bit-exact)
# ------------------------------------------------------------------------------------------------
def transpose(x: np.ndarray, perm: Sequence[int]) -> np.ndarray:
    """reference: src/kernels/cuda/transpose.cu:10-24"""
    return np.ascontiguousarray(np.transpose(x, perm))


def gather(data: np.ndarray, indices: np.ndarray, axis: int) -> np.ndarray:
    """reference: include/cuda/gather.h:33-55, src/kernels/cuda/gather.cu:4-29 (negative index wraps)"""
    idx = np.asarray(indices).astype(np.int64)
    idx = np.where(idx < 0, idx + data.shape[axis], idx)
    return np.take(data, idx, axis=axis)


def where(x: np.ndarray, y: np.ndarray, cond: np.ndarray) -> np.ndarray:
    """reference: src/kernels/cuda/where.cu:4-19 (cond ? x : y, 3-way broadcast)"""
    return np.where(np.asarray(cond).astype(bool), x, y)


def concat(xs: Sequence[np.ndarray], axis: int) -> np.ndarray:
    """reference: src/kernels/cuda/split_concat.cu:29-82"""
    return np.concatenate(xs, axis=axis)


def split(x: np.ndarray, axis: int, sizes: Sequence[int]) -> list[np.ndarray]:
    idx = np.cumsum(sizes)[:-1]
    return [np.ascontiguousarray(t) for t in np.split(x, idx, axis=axis)]


def slice_(x: np.ndarray, starts, ends, axes=None, steps=None) -> np.ndarray:
    """reference: src/operators/slice.cc (ONNX Slice); the CUDA kernel ignores steps
    (pad_slice.cc:35-40) — steps are honoured here."""
    axes = list(range(len(starts))) if axes is None else list(axes)
    steps = [1] * len(starts) if steps is None else list(steps)
    sl = [slice(None)] * x.ndim
    for a, s, e, st in zip(axes, starts, ends, steps):
        sl[a] = slice(s, e, st)
    return np.ascontiguousarray(x[tuple(sl)])


def pad(x: np.ndarray, pads: Sequence[int]) -> np.ndarray:
    """constant-0 pad; pads = [begin_0..begin_{r-1}, end_0..end_{r-1}] (reference: pad_slice.cu:25-101)"""
    r = x.ndim
    return np.pad(x, [(pads[i], pads[i + r]) for i in range(r)])


def expand(x: np.ndarray, shape: Sequence[int]) -> np.ndarray:
    """reference: src/kernels/cuda/expand.cu:10-49"""
    return np.ascontiguousarray(np.broadcast_to(x, shape))


# RoPE, rotate-half form, theta = 10000 (reference: src/kernels/cuda/rope.cu:6-31; the reference
# launch covers one (batch,pos) only (:85) — the definition below is the intended semantics)
def rope(pos: np.ndarray, x: np.ndarray, dim_head: int, theta: float = 10000.0) -> np.ndarray:
    X = np.asarray(x, dtype=np.float64)  # [..., seq, dim_model]
    P = np.asarray(pos, dtype=np.float64).reshape(X.shape[:-1])[..., None]
    dm = X.shape[-1]
    half = dim_head // 2
    j = np.arange(dm)
    jh = j % dim_head
    freq = theta ** (-(2.0 * (jh % half)) / dim_head)
    ang = P * freq
    # a trailing partial head (dm % dim_head != 0): its missing partner columns count as 0 — what the reference's own
    # test relies on (test_cuda_rope.cc:17-31: dim_model 32, head dim 128, expected values are pure cosines)
    pad = (-dm) % dim_head
    Xp = np.concatenate([X, np.zeros(X.shape[:-1] + (pad,))], axis=-1) if pad else X
    Xh = Xp.reshape(X.shape[:-1] + ((dm + pad) // dim_head, dim_head))
    rot = np.concatenate([-Xh[..., half:], Xh[..., :half]], axis=-1).reshape(Xp.shape)[..., :dm]
    return X * np.cos(ang) + rot * np.sin(ang)


# LRN across channels — ONNX LRN-13 (reference operator: src/operators/lrn.cc; front-end onnx.py:1101-1115 passes alpha, beta,
# bias, size through; the only kernel in the reference is Cambricon's cnnlLrn_v2, src/kernels/bang/lrn.cc:6-56, and no reference
# test asserts LRN values: PARITY UNPINNED for this op — the oracle is the ONNX operator definition).
def lrn(x: np.ndarray, size: int, alpha: float = 1e-4, beta: float = 0.75, bias: float = 1.0) -> np.ndarray:
    X = np.asarray(x, dtype=np.float64)
    C = X.shape[1]
    sq = X * X
    out = np.empty_like(X)
    lo, hi = (size - 1) // 2, size - 1 - (size - 1) // 2
    for c in range(C):
        a, b = max(0, c - lo), min(C - 1, c + hi)
        out[:, c] = X[:, c] / (bias + alpha / size * sq[:, a:b + 1].sum(axis=1)) ** beta
    return out


# Scaled-dot-product attention = the unfused chain MatMul -> Div/Mul -> (+mask) -> Softmax -> MatMul
def attention(q, k, v, scale: float, mask=None, causal: bool = False) -> np.ndarray:
    Q, K, V = (np.asarray(t, dtype=np.float64) for t in (q, k, v))
    s = np.matmul(Q, np.swapaxes(K, -1, -2)) * scale
    if mask is not None:
        s = s + np.asarray(mask, dtype=np.float64)
    if causal:
        sq, sk = s.shape[-2:]
        s = np.where(np.tril(np.ones((sq, sk), dtype=bool), k=sk - sq), s, -np.inf)
    return np.matmul(softmax(s, -1), V)


# AllReduce (reference: src/kernels/cuda/all_reduce.cc:10-33 — whole tensor, op in sum/prod/min/max/avg)
def all_reduce(kind: str, xs: Sequence[np.ndarray]) -> np.ndarray:
    S = np.stack([np.asarray(x, dtype=np.float64) for x in xs], 0)
    return {"sum": S.sum(0), "prod": S.prod(0), "min": S.min(0), "max": S.max(0), "avg": S.mean(0)}[kind]


# AttentionKVCache (reference: src/kernels/cuda/attention_kvcache.cu:8-169 — n = position_id[0] + 1 keys, the newest
# one taken from k / v and appended to the caches in place, scores / sqrt(D), softmax, times V)
def attention_kvcache(k_cache, v_cache, q, k, v, position: int):
    KC, VC = np.array(k_cache, dtype=np.float64), np.array(v_cache, dtype=np.float64)
    Q, K, V = (np.asarray(t, dtype=np.float64) for t in (q, k, v))
    n = int(position) + 1
    KC[:, :, n - 1, :] = K[:, :, 0, :]
    VC[:, :, n - 1, :] = V[:, :, 0, :]
    s = np.einsum("bhd,bhnd->bhn", Q[:, :, 0, :], KC[:, :, :n, :]) / np.sqrt(Q.shape[-1])
    p = softmax(s, -1)
    return np.einsum("bhn,bhnd->bhd", p, VC[:, :, :n, :])[:, :, None, :], KC, VC


# GatherElements (reference: src/kernels/cuda/gather_elements.cu:4-35; ONNX GatherElements-13)
def gather_elements(data: np.ndarray, indices: np.ndarray, axis: int) -> np.ndarray:
    d = np.asarray(data)
    idx = np.asarray(indices).astype(np.int64)
    idx = np.where(idx < 0, idx + d.shape[axis], idx)
    sl = tuple(slice(0, n) if a != axis % d.ndim else slice(None) for a, n in enumerate(idx.shape))
    return np.take_along_axis(d[sl], idx, axis)


# DepthToSpace (reference: src/operators/transpose.cc:68-110 — reshape, transpose {0,3,4,1,5,2} (DCR) or
# {0,1,4,2,5,3} (CRD), reshape)
def depth_to_space(x: np.ndarray, blocksize: int, mode: str = "DCR") -> np.ndarray:
    n, c, h, w = x.shape
    b = blocksize
    if mode == "DCR":
        t = x.reshape(n, b, b, c // (b * b), h, w).transpose(0, 3, 4, 1, 5, 2)
    else:
        t = x.reshape(n, c // (b * b), b, b, h, w).transpose(0, 1, 4, 2, 5, 3)
    return t.reshape(n, c // (b * b), h * b, w * b)


# Extend (reference: src/operators/extend.cc:14-18, src/kernels/cuda/extend.cu:3-15): num + 1 copies along dim
def extend(x: np.ndarray, dim: int, num: int) -> np.ndarray:
    return np.concatenate([np.asarray(x)] * (num + 1), axis=dim)


# Resize (reference: src/kernels/cuda/resize.cu:6-196 — coordinate transforms :20-49, nearest rounding :7-17,
# clamped even neighbours :111-116, linear / cubic (A = -0.75) coefficients :118-134, separable product :139-190).
# nearest rounding follows ONNX (round_prefer_floor = ceil(x - 0.5)); the reference's floor(x + 0.4) agrees wherever
# the fractional part is not in (0.5, 0.6) — every reference test.
def resize(x: np.ndarray, out_shape, scales, mode: str = "nearest", coord_mode: str = "half_pixel",
           nearest_mode: str = "round_prefer_floor", roi=None) -> np.ndarray:
    X = np.asarray(x, dtype=np.float64)
    nd = X.ndim
    out = np.zeros(tuple(out_shape), dtype=np.float64)

    def src(i, d):
        s = np.float32(scales[d])
        n_in = X.shape[d]
        length = np.float32(s * np.float32(n_in))
        if coord_mode == "half_pixel":
            return (i + 0.5) / s - 0.5
        if coord_mode == "pytorch_half_pixel":
            return (i + 0.5) / s - 0.5 if length > 1 else 0.0
        if coord_mode == "align_corners":
            return 0.0 if length == 1 else i * (n_in - 1) / (length - 1)
        if coord_mode == "asymmetric":
            return i / s
        li = int(length)
        rs, re = roi[d], roi[d + nd]
        return rs * (n_in - 1) + i * (re - rs) * (n_in - 1) / (li - 1) if li > 1 else 0.5 * (rs + re) * (n_in - 1)

    def coef(r):
        if mode == "linear":
            return [1 - r, r]
        A = -0.75
        return [((A * (r + 1) - 5 * A) * (r + 1) + 8 * A) * (r + 1) - 4 * A, ((A + 2) * r - (A + 3)) * r * r + 1,
                ((A + 2) * (1 - r) - (A + 3)) * (1 - r) * (1 - r) + 1,
                ((A * ((1 - r) + 1) - 5 * A) * ((1 - r) + 1) + 8 * A) * ((1 - r) + 1) - 4 * A]

    rnd = {"round_prefer_floor": lambda c: math.ceil(c - 0.5), "round_prefer_ceil": lambda c: math.floor(c + 0.5),
           "floor": math.floor, "ceil": math.ceil}[nearest_mode]
    for idx in np.ndindex(*out.shape):
        if mode == "nearest":
            sidx = tuple(i if X.shape[d] == out.shape[d] else min(max(rnd(src(i, d)), 0), X.shape[d] - 1)
                         for d, i in enumerate(idx))
            out[idx] = X[sidx]
            continue
        terms = [((), 1.0)]
        for d, i in enumerate(idx):
            if X.shape[d] == out.shape[d]:
                terms = [(t + (i,), w) for t, w in terms]
                continue
            c = src(i, d)
            fl = math.floor(c)
            ws = coef(c - fl)
            n = len(ws)
            nb = [min(max(fl - n // 2 + 1 + j, 0), X.shape[d] - 1) for j in range(n)]
            terms = [(t + (nb[j],), w * ws[j]) for t, w in terms for j in range(n)]
        out[idx] = sum(X[t] * w for t, w in terms)
    return out


# ConvTranspose2d (reference: src/operators/conv.cc:252-268 shape rule; cuDNN backward-data semantics,
# src/kernels/cuda/conv_transposed.cc:46-230): scatter form — every input element adds x * w[f, c, :, :] at
# (iy * sh - ph + r * dh, ix * sw - pw + s * dw).
def conv_transpose2d(x, w, ph=0, pw=0, sh=1, sw=1, dh=1, dw=1, oph=0, opw=0, groups=1) -> np.ndarray:
    X, W = np.asarray(x, dtype=np.float64), np.asarray(w, dtype=np.float64)
    n, f, h, wd = X.shape
    f2, cg, r, s = W.shape
    assert f == f2
    oh = (h - 1) * sh - 2 * ph + dh * (r - 1) + oph + 1
    ow = (wd - 1) * sw - 2 * pw + dw * (s - 1) + opw + 1
    fg = f // groups
    full = np.zeros((n, cg * groups, oh + 2 * ph + dh * r, ow + 2 * pw + dw * s))
    for g in range(groups):
        for fl in range(fg):
            ff = g * fg + fl
            for rr in range(r):
                for ss in range(s):
                    contrib = X[:, ff, :, :, None] * W[ff, :, rr, ss]  # [n, h, w, cg]
                    full[:, g * cg:(g + 1) * cg, rr * dh:rr * dh + (h - 1) * sh + 1:sh,
                         ss * dw:ss * dw + (wd - 1) * sw + 1:sw] += np.moveaxis(contrib, -1, 1)
    return full[:, :, ph:ph + oh, pw:pw + ow]
