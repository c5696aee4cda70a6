"""TensorProto <-> numpy (the two functions the front-end uses: onnx.py:26)."""
from __future__ import annotations

import numpy as np

from . import TensorProto

_NP = {TensorProto.FLOAT: np.float32, TensorProto.UINT8: np.uint8, TensorProto.INT8: np.int8, TensorProto.UINT16: np.uint16,
       TensorProto.INT16: np.int16, TensorProto.INT32: np.int32, TensorProto.INT64: np.int64, TensorProto.BOOL: np.bool_,
       TensorProto.FLOAT16: np.float16, TensorProto.DOUBLE: np.float64, TensorProto.UINT32: np.uint32, TensorProto.UINT64: np.uint64}
_ONNX = {np.dtype(v): k for k, v in _NP.items()}


def to_array(tensor) -> np.ndarray:
    dt = _NP[tensor.data_type]
    dims = tuple(tensor.dims)
    if tensor.HasField("raw_data"):
        return np.frombuffer(tensor.raw_data, dtype=np.dtype(dt).newbyteorder("<")).astype(dt).reshape(dims)
    if tensor.data_type == TensorProto.FLOAT:
        return np.asarray(tensor.float_data, dtype=dt).reshape(dims)
    if tensor.data_type == TensorProto.DOUBLE:
        return np.asarray(tensor.double_data, dtype=dt).reshape(dims)
    if tensor.data_type == TensorProto.INT64:
        return np.asarray(tensor.int64_data, dtype=dt).reshape(dims)
    if tensor.data_type in (TensorProto.UINT32, TensorProto.UINT64):
        return np.asarray(tensor.uint64_data, dtype=dt).reshape(dims)
    if tensor.data_type == TensorProto.FLOAT16:  # stored as the bit patterns in int32_data
        return np.asarray(tensor.int32_data, dtype=np.uint16).view(np.float16).reshape(dims)
    return np.asarray(tensor.int32_data, dtype=np.int32).astype(dt).reshape(dims)


def from_array(arr: np.ndarray, name: str | None = None):
    arr = np.asarray(arr)
    t = TensorProto()
    t.dims.extend(arr.shape)
    if name:
        t.name = name
    t.data_type = _ONNX[arr.dtype]
    t.raw_data = np.ascontiguousarray(arr).astype(arr.dtype.newbyteorder("<")).tobytes()
    return t
