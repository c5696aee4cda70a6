"""Host-side operator glue over the C ABI — the Python mirror of the reference's
`Kernel::compute(op, runtime)` classes: take dense device tensors + op attributes, do the shape
inference of the reference op class, allocate the output and enqueue the HIP kernel.

torch is used ONLY as the device-memory container (data_ptr / empty); no torch op computes a result
here. Every function takes the `RocmRuntime` first, like `compute(op, context)`.

Reference glue these mirror (file:line cited per function):
  matmul      src/kernels/cuda/matmul.cc:67-174        + src/operators/matmul.cc:26-49
  softmax     src/kernels/cuda/softmax.cc:9-31         + src/operators/softmax.cc
  layer_norm  src/kernels/cuda/layer_norm.cc:9-58      + src/operators/layer_norm.cc:5-32
  binary      src/kernels/cuda/element_wise.cc:13-175  + src/utils/operator_utils.cc:6-32
  unary/cast  src/kernels/cuda/unary.cc:30-122
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Sequence

import torch

from ._lib import check, lib
from .runtime import DType, RocmRuntime

_TORCH2DT = {
    torch.float32: DType.F32,
    torch.float16: DType.F16,
    torch.bfloat16: DType.BF16,
    torch.float64: DType.F64,
    torch.int8: DType.I8,
    torch.uint8: DType.U8,
    torch.int16: DType.I16,
    torch.int32: DType.I32,
    torch.int64: DType.I64,
    torch.bool: DType.BOOL,
}
_DT2TORCH = {v: k for k, v in _TORCH2DT.items()}
_DT2TORCH[DType.U32] = torch.int32  # storage only


def dtype_of(t: torch.Tensor) -> int:
    try:
        return int(_TORCH2DT[t.dtype])
    except KeyError:
        raise TypeError(f"unsupported tensor dtype {t.dtype}") from None


def torch_dtype(dt: int) -> torch.dtype:
    return _DT2TORCH[DType(dt)]


def _ptr(t: torch.Tensor | None) -> C.c_void_p:
    if t is None:
        return C.c_void_p(0)
    if not t.is_contiguous():
        raise ValueError("tensors must be dense, contiguous, row-major (reference: tensor.cc:74-82)")
    if t.device.type != "cuda":
        raise ValueError("tensor is not in device memory")
    return C.c_void_p(t.data_ptr())


def _i64arr(vals: Sequence[int]):
    return (C.c_int64 * max(len(vals), 1))(*vals)


def _i32arr(vals: Sequence[int]):
    return (C.c_int * max(len(vals), 1))(*vals)


# ------------------------------------------------------------------------------------------------
# shape helpers (reference: infer_broadcast, src/utils/operator_utils.cc:6-32)
# ------------------------------------------------------------------------------------------------
def infer_broadcast(a: Sequence[int], b: Sequence[int]) -> list[int]:
    ra, rb = len(a), len(b)
    r = max(ra, rb)
    out = []
    for i in range(r):
        da = a[i - (r - ra)] if i >= r - ra else 1
        db = b[i - (r - rb)] if i >= r - rb else 1
        if da != db and da != 1 and db != 1:
            raise ValueError(f"shapes {list(a)} and {list(b)} are not broadcastable")
        out.append(max(da, db) if (da != 0 and db != 0) else 0)
    return out


def broadcast_strides(shape: Sequence[int], out_shape: Sequence[int]) -> list[int]:
    """Element strides of a dense tensor of `shape` viewed in `out_shape` (0 where broadcast) — computed by the library
    (infini_rocm_broadcast_strides, csrc/shaped.hip: the one implementation the plugin kernels use too)."""
    out = _i64arr([0] * len(out_shape))
    check(lib().infini_rocm_broadcast_strides(len(shape), _i64arr(list(shape)), len(out_shape), _i64arr(list(out_shape)), out))
    return list(out)[:len(out_shape)]


def _shape(t: torch.Tensor):
    """(rank, int64 array) of a tensor's dims, as the from-shape entry points take them"""
    return t.dim(), _i64arr(list(t.shape))


# ------------------------------------------------------------------------------------------------
# MatMul
# ------------------------------------------------------------------------------------------------
def matmul(rt: RocmRuntime, a: torch.Tensor, b: torch.Tensor, bias: torch.Tensor | None = None,
           trans_a: bool = False, trans_b: bool = False, act: int = 0,
           out: torch.Tensor | None = None, head_split: tuple[int, int] | None = None) -> torch.Tensor:
    """C = op(A) op(B) (+bias) with the reference's batch-broadcast rule.

    Shape inference: src/operators/matmul.cc:26-49. Batch strides: zero when the operand is
    rank-2 or its broadcast batch is 1 (src/kernels/cuda/matmul.cc:124-137).
    head_split = (seq, head_dim): the [m, n] result is stored as [m / seq, n / head_dim, seq, head_dim] — MatMul ->
    Reshape([B, S, H, D]) -> Transpose(0, 2, 1, 3) in the GEMM epilogue (infini_rocm_matmul_headsplit).
    """
    # batch / m / n / k, the operands' batch strides and the bias strides come from the library's plan (infini_rocm_matmul_plan: the
    # glue of matmul.cc:86-137, shared with the plugin's MatmulRocm); only the OUTPUT shape (the operator's shape inference,
    # src/operators/matmul.cc:26-49, which the plugin gets from the reference's MatmulObj) is derived here.
    if a.dim() < 2 or b.dim() < 2:
        raise ValueError("matmul operands must have rank >= 2")
    plan = _i64arr([0] * 9)
    check(lib().infini_rocm_matmul_plan(*_shape(a), *_shape(b), bias.dim() if bias is not None else -1,
                                        _i64arr(list(bias.shape)) if bias is not None else None, int(trans_a), int(trans_b), plan))
    m, n = plan[1], plan[2]
    batch_shape = infer_broadcast(list(a.shape[:-2]), list(b.shape[:-2]))
    seq = hd = 0
    final_shape = batch_shape + [m, n]
    if head_split is not None:
        seq, hd = (int(v) for v in head_split)
        if seq <= 0 or hd <= 0 or m % seq or n % hd or hd % 8:
            raise ValueError(f"head_split {head_split} does not tile m = {m}, n = {n} (head_dim % 8 == 0)")
        final_shape = batch_shape + [m // seq, n // hd, seq, hd]
    if out is None:
        out = torch.empty(final_shape, dtype=a.dtype, device=a.device)
    check(lib().infini_rocm_matmul_shaped(rt.handle, dtype_of(a), _ptr(a), *_shape(a), _ptr(b), *_shape(b), _ptr(bias),
                                          bias.dim() if bias is not None else 0, _i64arr(list(bias.shape)) if bias is not None else None,
                                          _ptr(out), int(trans_a), int(trans_b), int(act), seq, hd))
    return out


def set_matmul_compute_type(rt: RocmRuntime, compute_type: str) -> None:
    """MatmulObj::getComputeType() for fp32 MatMuls: "default" / "tf32" exact fp32 products, "bf16" / "fp16" 16-bit products with
    fp32 accumulation and output (infini_rocm_matmul_set_compute_type). Sticky until reset to "default"."""
    check(lib().infini_rocm_matmul_set_compute_type(rt.handle, {"default": 0, "tf32": 0, "bf16": 1, "fp16": 2}[compute_type]))


def set_matmul_variant(rt: RocmRuntime, variant: int) -> None:
    check(lib().infini_rocm_matmul_set_variant(rt.handle, int(variant)))


def conv_last_route(rt: RocmRuntime) -> str:
    """Which conv2d implementation the most recent call on this runtime launched (include/infini_rocm.h)."""
    import ctypes as C

    v = C.c_char_p()
    check(lib().infini_rocm_conv2d_last_route(rt.handle, C.byref(v)))
    return v.value.decode()


def matmul_last_variant(rt: RocmRuntime) -> str:
    """Name of the GEMM kernel variant the most recent matmul on this runtime launched ("none" before the first)."""
    import ctypes as C

    v = C.c_int(-1)
    check(lib().infini_rocm_matmul_last_variant(rt.handle, C.byref(v)))
    return lib().infini_rocm_matmul_variant_name(v.value).decode() if v.value >= 0 else "none"


def lrn(rt: RocmRuntime, x: torch.Tensor, size: int, alpha: float = 1e-4, beta: float = 0.75, bias: float = 1.0,
        out: torch.Tensor | None = None) -> torch.Tensor:
    """ONNX LRN across dim 1 of x [N, C, ...] (operators/lrn.h)."""
    if x.dim() < 2:
        raise ValueError("lrn expects [N, C, ...]")
    if out is None:
        out = torch.empty_like(x)
    n, c = x.shape[0], x.shape[1]
    inner = x.numel() // max(1, n * c)
    check(lib().infini_rocm_lrn(rt.handle, dtype_of(x), _ptr(x), _ptr(out), n, c, inner, int(size), float(alpha), float(beta),
                                float(bias)))
    return out


def set_conv_const_weights(rt: RocmRuntime, on: bool) -> None:
    """While on, conv2d treats its weights as constant data and caches their re-packed image (infini_rocm.h)."""
    check(lib().infini_rocm_conv2d_set_const_weights(rt.handle, 1 if on else 0))


def weight_cache_info(rt: RocmRuntime) -> dict:
    import ctypes as C

    n, b, e = C.c_size_t(), C.c_size_t(), C.c_uint64()
    check(lib().infini_rocm_weight_cache_info(rt.handle, C.byref(n), C.byref(b), C.byref(e)))
    return {"entries": n.value, "bytes": b.value, "epoch": e.value}


def set_conv_variant(rt: RocmRuntime, variant: int) -> None:
    """-1 heuristic, 1 generic implicit GEMM, 2 conv_s1 wherever eligible, 3 batched-GEMM route for pointwise, 4 = 2 without the patch
    kernel, 5 pointwise layers as one pixel-slot GEMM, 6 = 2 with the 8-wave patch kernel (include/infini_rocm.h)."""
    check(lib().infini_rocm_conv2d_set_variant(rt.handle, int(variant)))


def matmul_grouped(rt: RocmRuntime, a: torch.Tensor, ws: list, outs: list, biases: list | None = None, act: int = 0) -> list:
    """Several MatMuls of ONE left operand `a` [m, k] with separate weights `ws[j]` [k, n] (and row biases [n]) into separate
    outputs `outs[j]` [m, n], as one launch: batch index = member, zero A stride. The members' weights, biases and outputs
    must sit at uniform distances in memory (infini_rocm_matmul_grouped); raises ValueError otherwise."""
    g = len(ws)
    if g < 1 or len(outs) != g or (biases is not None and len(biases) != g):
        raise ValueError("matmul_grouped: one weight, output (and bias) per member")
    m, k = a.shape
    n = ws[0].shape[1]
    es = a.element_size()

    def stride(ts):
        if g == 1:
            return 0
        d = ts[1].data_ptr() - ts[0].data_ptr()
        if d % es or any(t.data_ptr() - ts[0].data_ptr() != j * d for j, t in enumerate(ts)):
            raise ValueError("matmul_grouped: members are not uniformly spaced in memory")
        return d // es

    for w, o in zip(ws, outs):
        if tuple(w.shape) != (k, n) or tuple(o.shape) != (m, n) or w.dtype != a.dtype or o.dtype != a.dtype:
            raise ValueError("matmul_grouped: members must share shapes and dtype")
    sb, sc = stride(ws), stride(outs)
    sbias = stride(biases) if biases is not None else 0
    check(lib().infini_rocm_matmul_grouped(rt.handle, dtype_of(a), _ptr(a), _ptr(ws[0]), _ptr(biases[0]) if biases is not None else None,
                                           _ptr(outs[0]), g, m, n, k, 0, 0, 0, sb, sc, sbias, 0, 1 if biases is not None else 0, int(act), 0, 0))
    return outs


def matmul_variants() -> list[str]:
    n = lib().infini_rocm_matmul_num_variants()
    return [lib().infini_rocm_matmul_variant_name(i).decode() for i in range(n)]


# ------------------------------------------------------------------------------------------------
# Softmax / LayerNorm / RMSNorm
# ------------------------------------------------------------------------------------------------
def _real_axis(axis: int, rank: int) -> int:  # reference: get_real_axis, operator_utils.cc
    if not -rank <= axis < rank:
        raise ValueError(f"axis {axis} out of range for rank {rank}")
    return axis % rank if rank else 0


def softmax(rt: RocmRuntime, x: torch.Tensor, axis: int, out: torch.Tensor | None = None) -> torch.Tensor:
    axis = _real_axis(axis, x.dim())
    dims = list(x.shape)
    outer = math.prod(dims[:axis])
    inner = math.prod(dims[axis + 1:])
    if out is None:
        out = torch.empty_like(x)
    check(lib().infini_rocm_softmax(rt.handle, dtype_of(x), _ptr(x), _ptr(out), outer, dims[axis], inner))
    return out


def layer_norm(rt: RocmRuntime, x: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor | None = None,
               eps: float = 1e-5, axis: int = -1, out: torch.Tensor | None = None) -> torch.Tensor:
    axis = _real_axis(axis, x.dim())
    dims = list(x.shape)
    outer = math.prod(dims[:axis])
    norm = math.prod(dims[axis:])
    if out is None:
        out = torch.empty_like(x)
    check(lib().infini_rocm_layer_norm(rt.handle, dtype_of(x), _ptr(x), _ptr(scale), _ptr(bias), _ptr(out),
                                       outer, norm, scale.numel(), bias.numel() if bias is not None else 0,
                                       float(eps)))
    return out


def rms_norm(rt: RocmRuntime, x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-5,
             out: torch.Tensor | None = None) -> torch.Tensor:
    dims = list(x.shape)
    if out is None:
        out = torch.empty_like(x)
    check(lib().infini_rocm_rms_norm(rt.handle, dtype_of(x), _ptr(x), _ptr(weight), _ptr(out),
                                     math.prod(dims[:-1]), dims[-1], float(eps)))
    return out


def attention(rt: RocmRuntime, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float | torch.Tensor,
              mask: torch.Tensor | None = None, causal: bool = False, scale_is_div: bool = False,
              out: torch.Tensor | None = None, head_merge: int = 0) -> torch.Tensor:
    """softmax(scale * q k^T + mask) v over [..., S, D] (leading dims = batch x heads). `mask`: additive, shape
    [G, Sk] (one row per key sequence) or [G, Sq, Sk] (one row per query: e.g. a causal mask passed as a tensor) with G
    dividing the number of (batch, head) pairs (row g serves pairs g*BH/G .. (g+1)*BH/G - 1).
    `scale`: a float, or a one-element device tensor (then multiply / divide per scale_is_div).
    head_merge = H > 0: the result is stored as [BH / H, Sq, H, D] (Transpose(0, 2, 1, 3) of the plain [BH / H, H, Sq, D])."""
    if q.dim() < 3 or k.shape != v.shape or q.shape[:-2] != k.shape[:-2] or q.shape[-1] != k.shape[-1]:
        raise ValueError("attention expects q [..., Sq, D] and k, v [..., Sk, D]")
    bh = 1
    for d in q.shape[:-2]:
        bh *= d
    sq, sk, hd = q.shape[-2], k.shape[-2], q.shape[-1]
    if head_merge and (head_merge < 0 or bh % head_merge):
        raise ValueError(f"head_merge {head_merge} does not divide batch x heads = {bh}")
    if out is None:
        out = torch.empty((bh // head_merge, sq, head_merge, hd), dtype=q.dtype, device=q.device) if head_merge else torch.empty_like(q)
    group = 1
    mask_2d = 0
    if mask is not None:
        if mask.dim() == 3:  # full additive mask [G, Sq, Sk]
            if mask.shape[1] != sq:
                raise ValueError("a rank-3 mask must be [G, Sq, Sk]")
            mask_2d = 1
        elif mask.dim() != 2:
            raise ValueError("mask must be [G, Sk] or [G, Sq, Sk]")
        if mask.shape[-1] != sk or bh % mask.shape[0] != 0 or mask.dtype != q.dtype:
            raise ValueError("mask must be [G, Sk] / [G, Sq, Sk] of q's dtype with G dividing batch x heads")
        group = bh // mask.shape[0]
    dev_scale = scale if isinstance(scale, torch.Tensor) else None
    check(lib().infini_rocm_attention_ex(rt.handle, dtype_of(q), _ptr(q), _ptr(k), _ptr(v), _ptr(mask), _ptr(out), bh, sq, sk,
                                         hd, group, _ptr(dev_scale), int(scale_is_div),
                                         0.0 if dev_scale is not None else float(scale), int(causal), int(head_merge), mask_2d))
    return out


def attention_kvcache(rt: RocmRuntime, k_cache: torch.Tensor, v_cache: torch.Tensor, q: torch.Tensor, k: torch.Tensor,
                      v: torch.Tensor, position_id: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """Decode step (operators/attention_kvcache.h): caches [B, H, max_seq, D] are appended IN PLACE at
    position_id[0]; q, k, v [B, H, 1, D]; returns [B, H, 1, D]."""
    if k_cache.dim() != 4 or k_cache.shape != v_cache.shape:
        raise ValueError("caches must be rank-4 [B, H, max_seq, D]")  # reference: IT_ASSERT(rank == 4)
    b, h, ms, d = k_cache.shape
    for t_ in (q, k, v):
        if tuple(t_.shape) != (b, h, 1, d) or t_.dtype != k_cache.dtype:
            raise ValueError("q, k, v must be [B, H, 1, D] of the caches' dtype")
    if out is None:
        out = torch.empty_like(q)
    check(lib().infini_rocm_attention_kvcache(rt.handle, dtype_of(q), _ptr(k_cache), _ptr(v_cache), _ptr(q), _ptr(k),
                                              _ptr(v), dtype_of(position_id), _ptr(position_id), _ptr(out), b * h, ms, d))
    return out


def add_layer_norm(rt: RocmRuntime, a: torch.Tensor, b: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor | None,
                   eps: float, rms: bool = False, out: torch.Tensor | None = None, pre: torch.Tensor | None = None) -> torch.Tensor:
    """LayerNorm (or RMSNorm) over the last dim of a + b in one pass (the sum is rounded like the chain's). `pre`: one row
    of n elements added to a first — (a + pre) + b, the MatMul -> Add(bias) -> Add(residual) -> Norm chain of the ONNX
    front-end (infini_rocm_bias_add_norm)."""
    if a.shape != b.shape or a.dtype != b.dtype:
        raise ValueError("add_layer_norm: operands must match")
    n = a.shape[-1]
    if pre is not None and (pre.numel() != n or pre.dtype != a.dtype):
        raise ValueError("add_layer_norm: pre must be one row of the normalised size")
    if out is None:
        out = torch.empty_like(a)
    check(lib().infini_rocm_bias_add_norm(rt.handle, dtype_of(a), int(rms), _ptr(a), _ptr(pre), _ptr(b), _ptr(scale), _ptr(bias),
                                          _ptr(out), a.numel() // n, n, scale.numel(), bias.numel() if bias is not None else 0,
                                          float(eps)))
    return out


def rope(rt: RocmRuntime, pos: torch.Tensor, x: torch.Tensor, dim_head: int = 128, theta: float = 10000.0,
         out: torch.Tensor | None = None, head_split: bool = False) -> torch.Tensor:
    """RoPE(pos [B, S], x [B, S, dim_model]) (operators/rope.h; dim_head 128 / theta 1e4 as rope.cc:25)."""
    if x.dim() != 3 or pos.dim() != 2 or tuple(pos.shape) != tuple(x.shape[:2]):
        raise ValueError("rope expects pos [B, S] and x [B, S, dim_model]")  # reference: IT_ASSERT(nDims == 3 ...)
    if head_split:  # the result as [B, H, S, D]: RoPE -> Reshape([B, S, H, D]) -> Transpose(0, 2, 1, 3) in one pass
        if x.shape[2] % dim_head:
            raise ValueError("rope: head_split needs whole heads")
        if out is None:
            out = torch.empty((x.shape[0], x.shape[2] // dim_head, x.shape[1], dim_head), dtype=x.dtype, device=x.device)
        check(lib().infini_rocm_rope_headsplit(rt.handle, dtype_of(x), dtype_of(pos), _ptr(pos), _ptr(x), _ptr(out),
                                               x.shape[0] * x.shape[1], x.shape[2], int(dim_head), float(theta), x.shape[1]))
        return out
    if out is None:
        out = torch.empty_like(x)
    check(lib().infini_rocm_rope(rt.handle, dtype_of(x), dtype_of(pos), _ptr(pos), _ptr(x), _ptr(out),
                                 x.shape[0] * x.shape[1], x.shape[2], int(dim_head), float(theta)))
    return out


# ------------------------------------------------------------------------------------------------
# Element-wise
# ------------------------------------------------------------------------------------------------
BINARY_OPS = {
    "add": 0, "sub": 1, "mul": 2, "div": 3, "pow": 4, "min": 5, "max": 6,
    "equal": 7, "greater": 8, "greater_equal": 9, "less": 10, "less_equal": 11, "add_relu": 12,
}
UNARY_OPS = {
    "relu": 0, "sigmoid": 1, "tanh": 2, "abs": 3, "sqrt": 4, "gelu": 5, "silu": 6, "neg": 7,
    "erf": 8, "hard_sigmoid": 9, "hard_swish": 10, "exp": 11, "log": 12, "reciprocal": 13,
    "elu": 14, "leaky_relu": 15, "clip": 16, "sin": 17, "cos": 18, "ceil": 19, "floor": 20,
    "round": 21,
}


def binary(rt: RocmRuntime, op: str, a: torch.Tensor, b: torch.Tensor,
           out: torch.Tensor | None = None) -> torch.Tensor:
    if a.dtype != b.dtype:
        raise TypeError("binary operands must have the same dtype")
    out_shape = infer_broadcast(list(a.shape), list(b.shape))
    if out is None:
        out = torch.empty(out_shape, dtype=a.dtype, device=a.device)
    check(lib().infini_rocm_binary_shaped(rt.handle, BINARY_OPS[op], dtype_of(a), _ptr(a), *_shape(a), _ptr(b), *_shape(b), _ptr(out),
                                          len(out_shape), _i64arr(out_shape)))
    return out


def bias_residual(rt: RocmRuntime, a: torch.Tensor, bias: torch.Tensor, residual: torch.Tensor, relu: bool = True,
                  out: torch.Tensor | None = None) -> torch.Tensor:
    """relu?(a + bias[c] + residual) for a [N, C, ...] tensor and a per-channel bias: the fused Add -> Add -> Relu tail."""
    if a.shape != residual.shape or bias.numel() != a.shape[1]:
        raise ValueError("bias_residual: a / residual [N, C, ...], bias [C]")
    if out is None:
        out = torch.empty_like(a)
    inner = 1
    for d in a.shape[2:]:
        inner *= d
    check(lib().infini_rocm_bias_residual(rt.handle, dtype_of(a), _ptr(a), _ptr(bias), _ptr(residual), _ptr(out), a.shape[0],
                                          a.shape[1], inner, int(relu)))
    return out


def unary(rt: RocmRuntime, op: str, x: torch.Tensor, p0: float = float("nan"), p1: float = float("nan"),
          out: torch.Tensor | None = None) -> torch.Tensor:
    if out is None:
        out = torch.empty_like(x)
    check(lib().infini_rocm_unary(rt.handle, UNARY_OPS[op], dtype_of(x), _ptr(x), _ptr(out), x.numel(),
                                  float(p0), float(p1)))
    return out


def silu_mul(rt: RocmRuntime, a: torch.Tensor, b: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """silu(a) * b in one pass (the Silu -> Mul pair of a gated MLP); bit-identical to unary("silu") followed by binary("mul")."""
    if a.shape != b.shape or a.dtype != b.dtype:
        raise ValueError("silu_mul: operands must share shape and dtype")
    if out is None:
        out = torch.empty_like(a)
    check(lib().infini_rocm_silu_mul(rt.handle, dtype_of(a), _ptr(a), _ptr(b), _ptr(out), a.numel()))
    return out


def cast(rt: RocmRuntime, x: torch.Tensor, dst: torch.dtype, out: torch.Tensor | None = None) -> torch.Tensor:
    if out is None:
        out = torch.empty(x.shape, dtype=dst, device=x.device)
    check(lib().infini_rocm_cast(rt.handle, dtype_of(x), int(_TORCH2DT[dst]), _ptr(x), _ptr(out), x.numel()))
    return out


# ------------------------------------------------------------------------------------------------
# Conv2d / Pool / BatchNorm / Reduce
# ------------------------------------------------------------------------------------------------
def conv2d(rt: RocmRuntime, x: torch.Tensor, w: torch.Tensor, ph: int = 0, pw: int = 0, sh: int = 1,
           sw: int = 1, dh: int = 1, dw: int = 1, bias: torch.Tensor | None = None, act: int = 0,
           out: torch.Tensor | None = None, residual: torch.Tensor | None = None) -> torch.Tensor:
    """NCHW x FCRS cross-correlation. Shape rule: src/operators/conv.cc:47-114 (groups = C / w.shape[1]);
    glue mirrored: src/kernels/cuda/conv.cc:57-168."""
    n, c, h, wd = x.shape
    f, cpg, r, s = w.shape
    if c % cpg != 0:
        raise ValueError("input channels not divisible by weight channels")  # reference: IT_ASSERT
    groups = c // cpg
    if f % groups != 0:
        raise ValueError("filters not divisible by groups")
    oh = (h - (r - sh) * dh + ph * 2) // sh
    ow = (wd - (s - sw) * dw + pw * 2) // sw
    if out is None:
        out = torch.empty((n, f, oh, ow), dtype=x.dtype, device=x.device)
    if residual is not None and (tuple(residual.shape) != tuple(out.shape) or residual.dtype != x.dtype):
        raise ValueError("residual must have the output's shape and dtype")
    check(lib().infini_rocm_conv2d_res(rt.handle, dtype_of(x), _ptr(x), _ptr(w), _ptr(bias), _ptr(residual), _ptr(out), n, c,
                                       h, wd, f, r, s, ph, pw, sh, sw, dh, dw, groups, int(act)))
    return out


def conv2d_pool(rt: RocmRuntime, x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None, ph: int, pw: int, sh: int, sw: int,
                pool_k: int, pool_s: int, pool_p: int, act: int = 1, out: torch.Tensor | None = None) -> torch.Tensor:
    """MaxPool(act(conv2d(x, w) + bias)) as one launch (infini_rocm_conv2d_pool: the 7 x 7 / 2 stem of a CNN); raises where the
    library does not serve the shape."""
    n, c, h, wd = x.shape
    f, cpg, r, s = w.shape
    oh, ow = (h + 2 * ph - r) // sh + 1, (wd + 2 * pw - s) // sw + 1
    p_h, p_w = (oh + 2 * pool_p - pool_k) // pool_s + 1, (ow + 2 * pool_p - pool_k) // pool_s + 1
    if out is None:
        out = torch.empty((n, f, p_h, p_w), dtype=x.dtype, device=x.device)
    check(lib().infini_rocm_conv2d_pool(rt.handle, dtype_of(x), _ptr(x), _ptr(w), _ptr(bias), _ptr(out), n, c, h, wd, f, r, s, ph, pw, sh, sw,
                                        1, 1, c // cpg, int(act), int(pool_k), int(pool_s), int(pool_p)))
    return out


def conv_transpose2d(rt: RocmRuntime, x: torch.Tensor, w: torch.Tensor, ph: int = 0, pw: int = 0, sh: int = 1, sw: int = 1,
                     dh: int = 1, dw: int = 1, oph: int = 0, opw: int = 0, groups: int = 1,
                     bias: torch.Tensor | None = None, act: int = 0) -> torch.Tensor:
    """x [N, F, H, W], w [F, C/g, R, S] -> [N, C, OH, OW] (src/operators/conv.cc:252-268)."""
    n, f, h, wd = x.shape
    f2, cg, r, s = w.shape
    if f != f2:
        raise ValueError("input channels != weight dim 0")  # reference: IT_ASSERT(f == weight->getDims()[0])
    oh = (h - 1) * sh - 2 * ph + dh * (r - 1) + oph + 1
    ow = (wd - 1) * sw - 2 * pw + dw * (s - 1) + opw + 1
    out = torch.empty((n, cg * groups, oh, ow), dtype=x.dtype, device=x.device)
    check(lib().infini_rocm_conv_transpose2d(rt.handle, dtype_of(x), _ptr(x), _ptr(w), _ptr(bias), _ptr(out), n, f, h, wd,
                                             cg, r, s, ph, pw, sh, sw, dh, dw, oph, opw, groups, int(act)))
    return out


def _pool(rt, kind, x, kh, kw, dh, dw, ph, pw, sh, sw, ceil_mode, out):
    rank3 = x.dim() == 3  # reference: rank-3 input is treated as H = 1 (src/operators/pooling.cc:10-13)
    n, c = x.shape[0], x.shape[1]
    h = 1 if rank3 else x.shape[2]
    w = x.shape[-1]

    def osz(i, k, d, p, s):
        v = (i + 2 * p - d * (k - 1) - 1) / s + 1
        return int(math.ceil(v) if ceil_mode else math.floor(v))

    oh, ow = osz(h, kh, dh, ph, sh), osz(w, kw, dw, pw, sw)
    if out is None:
        out = torch.empty((n, c, ow) if rank3 else (n, c, oh, ow), dtype=x.dtype, device=x.device)
    check(lib().infini_rocm_pool2d(rt.handle, kind, dtype_of(x), _ptr(x), _ptr(out), n, c, h, w, kh, kw, dh, dw,
                                   ph, pw, sh, sw, int(ceil_mode)))
    return out


def max_pool(rt, x, kh, kw, dh=1, dw=1, ph=0, pw=0, sh=1, sw=1, ceil_mode=0, out=None):
    return _pool(rt, 0, x, kh, kw, dh, dw, ph, pw, sh, sw, ceil_mode, out)


def avg_pool(rt, x, kh, kw, dh=1, dw=1, ph=0, pw=0, sh=1, sw=1, ceil_mode=0, out=None):
    return _pool(rt, 1, x, kh, kw, dh, dw, ph, pw, sh, sw, ceil_mode, out)


def batch_norm(rt: RocmRuntime, x: torch.Tensor, mean: torch.Tensor, var: torch.Tensor, scale: torch.Tensor,
               bias: torch.Tensor, eps: float = 1e-5, out: torch.Tensor | None = None) -> torch.Tensor:
    """Inference BN over dim 1; operator input order x, mean, var, scale, bias
    (include/operators/batch_norm.h:10-50); parameters are fp32 [C] (batch_norm.cc:13)."""
    for t in (mean, var, scale, bias):
        if t.dtype != torch.float32 or t.numel() != x.shape[1]:
            raise ValueError("batch_norm parameters must be float32 of shape [C]")
    if out is None:
        out = torch.empty_like(x)
    n, c = x.shape[0], x.shape[1]
    check(lib().infini_rocm_batch_norm(rt.handle, dtype_of(x), _ptr(x), _ptr(mean), _ptr(var), _ptr(scale),
                                       _ptr(bias), _ptr(out), n, c, math.prod(x.shape[2:]), float(eps)))
    return out


def reduce(rt: RocmRuntime, kind: str, x: torch.Tensor, axes: Sequence[int] | None = None, keepdims: bool = True,
           out: torch.Tensor | None = None) -> torch.Tensor:
    """ReduceSum / ReduceMean (src/operators/reduce.cc; kernel glue src/kernels/cuda/reduce.cc:10-108)."""
    rank = x.dim()
    ax = sorted({_real_axis(a, rank) for a in axes}) if axes else list(range(rank))
    flags = [1 if d in ax else 0 for d in range(rank)]
    oshape = [1 if flags[d] else x.shape[d] for d in range(rank)] if keepdims else \
        [x.shape[d] for d in range(rank) if not flags[d]]
    if out is None:
        out = torch.empty(oshape, dtype=x.dtype, device=x.device)
    check(lib().infini_rocm_reduce(rt.handle, {"sum": 0, "mean": 1}[kind], dtype_of(x), _ptr(x), _ptr(out), rank,
                                   _i64arr(list(x.shape)), _i32arr(flags)))
    return out


# ------------------------------------------------------------------------------------------------
# Data movement / indexing
# ------------------------------------------------------------------------------------------------
def transpose(rt: RocmRuntime, x: torch.Tensor, perm: Sequence[int], out: torch.Tensor | None = None) -> torch.Tensor:
    perm = [int(p) for p in perm]
    if sorted(perm) != list(range(x.dim())):
        raise ValueError("bad permutation")
    if out is None:
        out = torch.empty([x.shape[p] for p in perm], dtype=x.dtype, device=x.device)
    check(lib().infini_rocm_transpose(rt.handle, dtype_of(x), _ptr(x), _ptr(out), x.dim(), _i64arr(list(x.shape)),
                                      _i32arr(perm)))
    return out


def reshape(rt: RocmRuntime, x: torch.Tensor, shape: Sequence[int], out: torch.Tensor | None = None) -> torch.Tensor:
    """Reshape / Flatten / Identity / Squeeze / Unsqueeze: a device copy (reshape.cc:4-21; the planner
    never aliases input and output)."""
    if out is None:
        out = torch.empty(list(shape), dtype=x.dtype, device=x.device)
    if out.numel() != x.numel():
        raise ValueError("reshape changes the element count")
    check(lib().infini_rocm_copy_inside(rt.handle, _ptr(out), _ptr(x), x.numel() * x.element_size()))
    return out


def expand(rt: RocmRuntime, x: torch.Tensor, shape: Sequence[int], out: torch.Tensor | None = None) -> torch.Tensor:
    oshape = infer_broadcast(list(x.shape), list(shape))
    if out is None:
        out = torch.empty(oshape, dtype=x.dtype, device=x.device)
    check(lib().infini_rocm_expand_shaped(rt.handle, dtype_of(x), _ptr(x), *_shape(x), _ptr(out), len(oshape), _i64arr(oshape)))
    return out


def gather(rt: RocmRuntime, data: torch.Tensor, indices: torch.Tensor, axis: int = 0,
           out: torch.Tensor | None = None) -> torch.Tensor:
    """Output rank = data rank - 1 + index rank (include/operators/gather.h:27-49)."""
    axis = _real_axis(axis, data.dim())
    if indices.dtype not in (torch.int32, torch.int64):
        raise TypeError("gather indices must be int32 or int64")
    oshape = list(data.shape[:axis]) + list(indices.shape) + list(data.shape[axis + 1:])
    if out is None:
        out = torch.empty(oshape, dtype=data.dtype, device=data.device)
    check(lib().infini_rocm_gather_shaped(rt.handle, dtype_of(data), dtype_of(indices), _ptr(data), *_shape(data), _ptr(indices),
                                          indices.numel(), _ptr(out), axis))
    return out


def gather_elements(rt: RocmRuntime, data: torch.Tensor, indices: torch.Tensor, axis: int = 0,
                    out: torch.Tensor | None = None) -> torch.Tensor:
    """out[i] = data[i with coordinate `axis` replaced by indices[i]]; output has the index shape
    (src/operators/gather_elements.cc:27-39)."""
    axis = _real_axis(axis, data.dim())
    if indices.dtype not in (torch.int32, torch.int64):
        raise TypeError("gather_elements indices must be int32 or int64")
    if indices.dim() != data.dim():
        raise ValueError("data and indices must have the same rank")  # reference: checkShape
    if out is None:
        out = torch.empty(indices.shape, dtype=data.dtype, device=data.device)
    check(lib().infini_rocm_gather_elements(rt.handle, dtype_of(data), dtype_of(indices), _ptr(data), _ptr(indices),
                                            _ptr(out), data.dim(), _i64arr(list(data.shape)),
                                            _i64arr(list(indices.shape)), axis))
    return out


def depth_to_space(rt: RocmRuntime, x: torch.Tensor, blocksize: int, mode: str = "DCR") -> torch.Tensor:
    """ONNX DepthToSpace as reshape -> transpose -> reshape (src/operators/transpose.cc:68-110,
    src/kernels/cuda/transpose.cc:47-90: perm {0,3,4,1,5,2} for DCR, {0,1,4,2,5,3} for CRD)."""
    n, c, h, w = x.shape
    b = int(blocksize)
    if c % (b * b):
        raise ValueError("channels not divisible by blocksize^2")
    if mode == "DCR":
        y = transpose(rt, x.view(n, b, b, c // (b * b), h, w), (0, 3, 4, 1, 5, 2))
    else:
        y = transpose(rt, x.view(n, c // (b * b), b, b, h, w), (0, 1, 4, 2, 5, 3))
    return y.view(n, c // (b * b), h * b, w * b)


def extend(rt: RocmRuntime, x: torch.Tensor, dim: int, num: int) -> torch.Tensor:
    """Repeat x (num + 1) times along `dim` (src/operators/extend.cc:14-18, src/kernels/cuda/extend.cu:3-15)."""
    dim = _real_axis(dim, x.dim())
    outer = math.prod(x.shape[:dim])
    blk = math.prod(x.shape[dim:])
    y = expand(rt, x.reshape(outer, 1, blk), (outer, num + 1, blk))
    shape = list(x.shape)
    shape[dim] *= num + 1
    return y.view(shape)


RESIZE_MODES = {"nearest": 0, "linear": 1, "cubic": 2}
RESIZE_COORD = {"half_pixel": 0, "pytorch_half_pixel": 1, "align_corners": 2, "asymmetric": 3, "tf_crop_and_resize": 4}
RESIZE_NEAREST = {"round_prefer_floor": 0, "round_prefer_ceil": 1, "floor": 2, "ceil": 3}


def resize(rt: RocmRuntime, x: torch.Tensor, out_shape: Sequence[int], scales: Sequence[float] | None = None,
           mode: str = "nearest", coord_mode: str = "half_pixel", nearest_mode: str = "round_prefer_floor",
           roi: Sequence[float] | None = None) -> torch.Tensor:
    """ONNX Resize on a dense tensor. `scales` defaults to out / in per dim (ResizeObj with the stretch policy,
    src/operators/resize.cc:117-123)."""
    if len(out_shape) != x.dim():
        raise ValueError("resize keeps the rank")
    if scales is None:
        scales = [o / i for o, i in zip(out_shape, x.shape)]
    out = torch.empty(list(out_shape), dtype=x.dtype, device=x.device)
    fs = (C.c_float * x.dim())(*[float(v) for v in scales])
    fr = (C.c_float * (2 * x.dim()))(*[float(v) for v in roi]) if roi is not None else None
    check(lib().infini_rocm_resize(rt.handle, dtype_of(x), _ptr(x), _ptr(out), x.dim(), _i64arr(list(x.shape)),
                                   _i64arr(list(out_shape)), fs, fr, RESIZE_MODES[mode], RESIZE_COORD[coord_mode],
                                   RESIZE_NEAREST[nearest_mode]))
    return out


def where(rt: RocmRuntime, x: torch.Tensor, y: torch.Tensor, cond: torch.Tensor,
          out: torch.Tensor | None = None) -> torch.Tensor:
    """cond ? x : y; operator input order x, y, cond (include/operators/where.h:9-34)."""
    oshape = infer_broadcast(infer_broadcast(list(x.shape), list(y.shape)), list(cond.shape))
    if out is None:
        out = torch.empty(oshape, dtype=x.dtype, device=x.device)
    check(lib().infini_rocm_where_shaped(rt.handle, dtype_of(x), dtype_of(cond), _ptr(x), *_shape(x), _ptr(y), *_shape(y), _ptr(cond),
                                         *_shape(cond), _ptr(out), len(oshape), _i64arr(oshape)))
    return out


def concat(rt: RocmRuntime, xs: Sequence[torch.Tensor], axis: int, out: torch.Tensor | None = None) -> torch.Tensor:
    axis = _real_axis(axis, xs[0].dim())
    oshape = list(xs[0].shape)
    oshape[axis] = sum(t.shape[axis] for t in xs)
    if out is None:
        out = torch.empty(oshape, dtype=xs[0].dtype, device=xs[0].device)
    n = len(xs)  # every input as one segment of ONE launch (infini_rocm_concat_shaped -> infini_rocm_strided_copy_multi)
    check(lib().infini_rocm_concat_shaped(rt.handle, out.element_size(), n, (C.c_void_p * n)(*[t.data_ptr() for t in xs]),
                                          _i64arr([t.shape[axis] for t in xs]), _ptr(out), len(oshape), _i64arr(oshape), axis))
    return out


def split(rt: RocmRuntime, x: torch.Tensor, axis: int, sizes: Sequence[int]) -> list[torch.Tensor]:
    axis = _real_axis(axis, x.dim())
    if sum(sizes) != x.shape[axis]:
        raise ValueError("split sizes do not add up")
    outs = []
    for sz in sizes:
        shp = list(x.shape)
        shp[axis] = sz
        outs.append(torch.empty(shp, dtype=x.dtype, device=x.device))
    n = len(outs)
    check(lib().infini_rocm_split_shaped(rt.handle, x.element_size(), n, (C.c_void_p * n)(*[o.data_ptr() for o in outs]),
                                         _i64arr(list(sizes)), _ptr(x), *_shape(x), axis))
    return outs


def slice_(rt: RocmRuntime, x: torch.Tensor, starts: Sequence[int], ends: Sequence[int],
           axes: Sequence[int] | None = None, steps: Sequence[int] | None = None) -> torch.Tensor:
    """ONNX Slice with positive steps (src/operators/slice.cc normalises starts/ends)."""
    rank = x.dim()
    axes = list(range(len(starts))) if axes is None else [_real_axis(a, rank) for a in axes]
    steps = [1] * len(starts) if steps is None else list(steps)
    st, sp, oshape = [0] * rank, [1] * rank, list(x.shape)
    for a, s, e, k in zip(axes, starts, ends, steps):
        d = x.shape[a]
        if k <= 0:
            raise ValueError("only positive steps are supported")
        s = min(max(s + d if s < 0 else s, 0), d)
        e = min(max(e + d if e < 0 else e, 0), d)
        st[a], sp[a], oshape[a] = s, k, max(0, -(-(e - s) // k))
    out = torch.empty(oshape, dtype=x.dtype, device=x.device)
    check(lib().infini_rocm_pad_slice(rt.handle, dtype_of(x), _ptr(x), _ptr(out), rank, _i64arr(list(x.shape)),
                                      _i64arr(oshape), _i64arr(st), _i64arr(sp), 0))
    return out


def pad(rt: RocmRuntime, x: torch.Tensor, pads: Sequence[int]) -> torch.Tensor:
    """Constant-0 pad; pads = [begin_0..begin_{r-1}, end_0..end_{r-1}] (include/operators/pad.h)."""
    rank = x.dim()
    if len(pads) != 2 * rank:
        raise ValueError("pads must have 2*rank entries")
    oshape = [x.shape[d] + pads[d] + pads[d + rank] for d in range(rank)]
    out = torch.empty(oshape, dtype=x.dtype, device=x.device)
    check(lib().infini_rocm_pad_shaped(rt.handle, dtype_of(x), _ptr(x), _ptr(out), rank, _i64arr(list(x.shape)), _i64arr(list(pads))))
    return out


# ------------------------------------------------------------------------------------------------
# Collectives (operators AllReduce{Sum,Prod,Min,Max,Avg}, AllGather, Broadcast, Send, Recv)
# ------------------------------------------------------------------------------------------------
_RED = {"sum": 0, "prod": 1, "min": 2, "max": 3, "avg": 4}


def all_reduce(rt: RocmRuntime, kind: str, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    if out is None:
        out = torch.empty_like(x)
    check(lib().infini_rocm_all_reduce(rt.handle, _RED[kind], dtype_of(x), _ptr(x), _ptr(out), x.numel()))
    return out


def all_reduce_async(rt: RocmRuntime, kind: str, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """The all-reduce on the runtime's comm stream, ordered after the work enqueued so far; later work on the runtime stream
    does not wait for it until comm_join (infini_rocm_all_reduce_async)."""
    if out is None:
        out = torch.empty_like(x)
    check(lib().infini_rocm_all_reduce_async(rt.handle, _RED[kind], dtype_of(x), _ptr(x), _ptr(out), x.numel()))
    return out


def comm_join(rt: RocmRuntime) -> None:
    check(lib().infini_rocm_comm_join(rt.handle))


def reduce_scatter(rt: RocmRuntime, x: torch.Tensor, direct: bool = False, out: torch.Tensor | None = None) -> torch.Tensor:
    """out = sum over ranks of x[rank] for x of shape [world, ...] (infini_rocm_reduce_scatter); direct: the one-hop xGMI
    exchange (grouped send / recv + local sum) instead of RCCL's algorithm choice."""
    world, _ = rt.comm_info()
    if x.shape[0] != world:
        raise ValueError("reduce_scatter: leading dim must be the world size")
    if out is None:
        out = torch.empty(x.shape[1:], dtype=x.dtype, device=x.device)
    check(lib().infini_rocm_reduce_scatter(rt.handle, dtype_of(x), _ptr(x), _ptr(out), out.numel(), 1 if direct else 0))
    return out


def all_gather(rt: RocmRuntime, x: torch.Tensor) -> list[torch.Tensor]:
    world, _ = rt.comm_info()
    buf = torch.empty((world,) + tuple(x.shape), dtype=x.dtype, device=x.device)
    check(lib().infini_rocm_all_gather(rt.handle, dtype_of(x), _ptr(x), _ptr(buf), x.numel()))
    return [buf[i] for i in range(world)]


def broadcast(rt: RocmRuntime, x: torch.Tensor, root: int, out: torch.Tensor | None = None) -> torch.Tensor:
    if out is None:
        out = torch.empty_like(x)
    check(lib().infini_rocm_broadcast(rt.handle, dtype_of(x), _ptr(x), _ptr(out), x.numel(), int(root)))
    return out


def send(rt: RocmRuntime, x: torch.Tensor, peer: int) -> None:
    check(lib().infini_rocm_send(rt.handle, dtype_of(x), _ptr(x), x.numel(), int(peer)))


def recv(rt: RocmRuntime, shape: Sequence[int], dtype: torch.dtype, peer: int, device=None) -> torch.Tensor:
    out = torch.empty(list(shape), dtype=dtype, device=device or f"cuda:{rt.device}")
    check(lib().infini_rocm_recv(rt.handle, int(_TORCH2DT[dtype]), _ptr(out), out.numel(), int(peer)))
    return out
