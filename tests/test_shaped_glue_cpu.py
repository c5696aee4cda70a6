"""The shape -> stride glue below the C ABI (csrc/shaped.hip), shared by the plugin kernels and the ctypes mirror: its pure parts
(infini_rocm_broadcast_strides, infini_rocm_matmul_plan) run without a GPU. Reference behaviour: infer_broadcast
(src/utils/operator_utils.cc:6-32), MatMul's batch / bias rules (src/kernels/cuda/matmul.cc:86-137)."""
import ctypes as C

import numpy as np
import pytest

from infinitensor_amd._lib import lib


def _arr(v):
    return (C.c_int64 * max(len(v), 1))(*v)


def bstrides(shape, out_shape):
    out = _arr([0] * len(out_shape))
    st = lib().infini_rocm_broadcast_strides(len(shape), _arr(shape), len(out_shape), _arr(out_shape), out)
    return st, list(out)[:len(out_shape)]


@pytest.mark.parametrize("shape,out_shape", [((3, 1, 5), (2, 3, 4, 5)), ((5,), (7, 5)), ((1,), (4, 4)), ((), (2, 3)), ((2, 3, 4), (2, 3, 4)),
                                              ((1, 1, 1), (6, 1, 2)), ((4, 1), (3, 4, 9)), ((0, 3), (0, 3))])
def test_broadcast_strides_match_numpy(shape, out_shape):
    st, got = bstrides(shape, out_shape)
    assert st == 0
    base = np.zeros(shape, dtype=np.int8)
    want = [s if d != 1 else 0 for s, d in zip(np.broadcast_to(base, out_shape).strides, out_shape)] if base.size else None
    if want is not None:
        # numpy reports stride 0 for broadcast dims; a dim of extent 1 in the OUTPUT may carry any stride
        got_n = [g if d != 1 else 0 for g, d in zip(got, out_shape)]
        assert got_n == want


@pytest.mark.parametrize("shape,out_shape", [((3,), (2, 4)), ((2, 3), (3,)), ((2, 1, 3), (2, 5, 4))])
def test_broadcast_strides_reject_what_does_not_broadcast(shape, out_shape):
    st, _ = bstrides(shape, out_shape)
    assert st != 0
    assert b"broadcast" in lib().infini_rocm_last_error()


def plan(a, b, bias=None, ta=0, tb=0):
    p = _arr([0] * 9)
    st = lib().infini_rocm_matmul_plan(len(a), _arr(a), len(b), _arr(b), len(bias) if bias is not None else -1,
                                       _arr(bias) if bias is not None else None, ta, tb, p)
    return st, list(p)


def test_matmul_plan_batch_broadcast_and_transposes():
    # [2, 3, 4, 5] x [5, 6]: B is shared by the 6 batches (zero stride), matmul.cc:124-137
    assert plan((2, 3, 4, 5), (5, 6)) == (0, [6, 4, 6, 5, 20, 0, 0, 0, 0])
    assert plan((4, 5), (7, 5, 6)) == (0, [7, 4, 6, 5, 0, 30, 0, 0, 0])
    assert plan((1, 4, 5), (3, 5, 6)) == (0, [3, 4, 6, 5, 0, 30, 0, 0, 0])
    assert plan((3, 5, 4), (3, 6, 5), ta=1, tb=1) == (0, [3, 4, 6, 5, 20, 30, 0, 0, 0])
    assert plan((4, 5), (5, 6)) == (0, [1, 4, 6, 5, 20, 30, 0, 0, 0])


def test_matmul_plan_bias_forms():
    assert plan((2, 4, 5), (5, 6), bias=(6,))[1][6:] == [0, 0, 1]          # one value per column
    assert plan((2, 4, 5), (5, 6), bias=(4, 1))[1][6:] == [0, 1, 0]        # one value per row
    assert plan((2, 4, 5), (5, 6), bias=(4, 6))[1][6:] == [0, 6, 1]        # a full [m, n] plane shared by the batches
    assert plan((2, 4, 5), (5, 6), bias=(2, 4, 6))[1][6:] == [24, 6, 1]    # per batch
    assert plan((2, 4, 5), (5, 6), bias=(2, 1, 6))[1][6:] == [6, 0, 1]     # per batch, one row
    assert plan((2, 4, 5), (5, 6), bias=(1,))[1][6:] == [0, 0, 0]          # a scalar


def test_matmul_plan_errors_are_the_references_asserts():
    assert plan((4, 5), (4, 6))[0] != 0 and b"K of A" in lib().infini_rocm_last_error()          # IT_ASSERT(kA == kB)
    assert plan((2, 4, 5), (3, 5, 6))[0] != 0                                                         # batch dims do not broadcast
    assert plan((2, 1, 4, 5), (1, 3, 5, 6))[0] != 0 and b"size-1 batch" in lib().infini_rocm_last_error()  # partial batch broadcast
    assert plan((5,), (5, 6))[0] != 0                                                                 # rank < 2
    assert plan((2, 3, 4, 5), (5, 6), bias=(3, 4, 6))[0] != 0                                         # partially broadcast bias batch
    assert plan((4, 5), (5, 6), bias=(7,))[0] != 0                                                    # bias does not broadcast
