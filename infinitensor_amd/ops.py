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
    """Element strides of a dense tensor of `shape` viewed in `out_shape` (0 where broadcast)."""
    r, ro = len(shape), len(out_shape)
    dense = [0] * r
    p = 1
    for i in range(r - 1, -1, -1):
        dense[i] = p
        p *= shape[i]
    out = [0] * ro
    for i in range(ro):
        j = i - (ro - r)
        if j >= 0 and shape[j] != 1:
            out[i] = dense[j]
    return out


# ------------------------------------------------------------------------------------------------
# MatMul
# ------------------------------------------------------------------------------------------------
def matmul(rt: RocmRuntime, a: torch.Tensor, b: torch.Tensor, bias: torch.Tensor | None = None,
           trans_a: bool = False, trans_b: bool = False, act: int = 0,
           out: torch.Tensor | None = None) -> torch.Tensor:
    """C = op(A) op(B) (+bias) with the reference's batch-broadcast rule.

    Shape inference: src/operators/matmul.cc:26-49. Batch strides: zero when the operand is
    rank-2 or its broadcast batch is 1 (src/kernels/cuda/matmul.cc:124-137).
    """
    sa, sb = list(a.shape), list(b.shape)
    if len(sa) < 2 or len(sb) < 2:
        raise ValueError("matmul operands must have rank >= 2")
    batch_shape = infer_broadcast(sa[:-2], sb[:-2])
    batch = math.prod(batch_shape) if batch_shape else 1
    k_a = sa[-2] if trans_a else sa[-1]
    k_b = sb[-1] if trans_b else sb[-2]
    if k_a != k_b:
        raise ValueError(f"matmul K mismatch: {k_a} vs {k_b}")  # reference: IT_ASSERT(kA == kB)
    m = sa[-1] if trans_a else sa[-2]
    n = sb[-2] if trans_b else sb[-1]
    k = k_a
    out_shape = batch_shape + [m, n]
    ba, bb = math.prod(sa[:-2]), math.prod(sb[:-2])
    if ba not in (1, batch) or bb not in (1, batch):
        raise ValueError("only full or size-1 batch broadcast is supported (reference matmul.cc:124-137)")
    stride_a = 0 if (ba == 1 and batch > 1) else m * k
    stride_b = 0 if (bb == 1 and batch > 1) else n * k
    if out is None:
        out = torch.empty(out_shape, dtype=a.dtype, device=a.device)
    bs_b = bs_m = bs_n = 0
    if bias is not None:
        st = broadcast_strides(list(bias.shape), out_shape)
        # collapse the batch dims of the bias into one stride (dense or broadcast)
        nb = len(out_shape) - 2
        bs_m, bs_n = st[-2], st[-1]
        lead = [s for s, d in zip(st[:nb], out_shape[:nb]) if d != 1]
        if all(s == 0 for s in lead):
            bs_b = 0
        else:
            bs_b = math.prod(bias.shape[-2:]) if bias.dim() >= 2 else 0
            if list(bias.shape[:-2]) and math.prod(bias.shape[:-2]) != batch:
                raise ValueError("partially-broadcast bias batch is not supported")
    check(lib().infini_rocm_matmul(rt.handle, dtype_of(a), _ptr(a), _ptr(b), _ptr(bias), _ptr(out),
                                   batch, m, n, k, int(trans_a), int(trans_b), stride_a, stride_b,
                                   bs_b, bs_m, bs_n, int(act)))
    return out


def set_matmul_variant(rt: RocmRuntime, variant: int) -> None:
    check(lib().infini_rocm_matmul_set_variant(rt.handle, int(variant)))


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


# ------------------------------------------------------------------------------------------------
# Element-wise
# ------------------------------------------------------------------------------------------------
BINARY_OPS = {
    "add": 0, "sub": 1, "mul": 2, "div": 3, "pow": 4, "min": 5, "max": 6,
    "equal": 7, "greater": 8, "greater_equal": 9, "less": 10, "less_equal": 11,
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
    sa = broadcast_strides(list(a.shape), out_shape)
    sb = broadcast_strides(list(b.shape), out_shape)
    if len(out_shape) > 8:
        raise ValueError("rank > 8 not supported")
    check(lib().infini_rocm_binary(rt.handle, BINARY_OPS[op], dtype_of(a), _ptr(a), _ptr(b), _ptr(out),
                                   len(out_shape), _i64arr(out_shape), _i64arr(sa), _i64arr(sb)))
    return out


def unary(rt: RocmRuntime, op: str, x: torch.Tensor, p0: float = float("nan"), p1: float = float("nan"),
          out: torch.Tensor | None = None) -> torch.Tensor:
    if out is None:
        out = torch.empty_like(x)
    check(lib().infini_rocm_unary(rt.handle, UNARY_OPS[op], dtype_of(x), _ptr(x), _ptr(out), x.numel(),
                                  float(p0), float(p1)))
    return out


def cast(rt: RocmRuntime, x: torch.Tensor, dst: torch.dtype, out: torch.Tensor | None = None) -> torch.Tensor:
    if out is None:
        out = torch.empty(x.shape, dtype=dst, device=x.device)
    check(lib().infini_rocm_cast(rt.handle, dtype_of(x), int(_TORCH2DT[dst]), _ptr(x), _ptr(out), x.numel()))
    return out
