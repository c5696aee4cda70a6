"""A minimal reader of the ONNX protobuf WIRE FORMAT (no `onnx` package in this image): ModelProto -> GraphProto -> NodeProto /
TensorProto / ValueInfoProto, only the fields the front-end-form tests need (op types, edges, int / ints / float attributes,
initializer names, shapes and small constant payloads). Field numbers from onnx.proto (ONNX IR v7+):
  ModelProto   1 ir_version, 7 graph, 8 opset_import {1 domain, 2 version}
  GraphProto   1 node, 2 name, 5 initializer, 11 input, 12 output
  NodeProto    1 input, 2 output, 3 name, 4 op_type, 5 attribute, 7 domain
  Attribute    1 name, 2 f, 3 i, 4 s, 5 t, 7 floats, 8 ints, 20 type
  TensorProto  1 dims, 2 data_type, 4 float_data, 5 int32_data, 7 int64_data, 8 name, 9 raw_data
  ValueInfo    1 name, 2 type {1 tensor_type {1 elem_type, 2 shape {1 dim {1 dim_value, 2 dim_param}}}}
Test infrastructure only (the product never reads ONNX: the reference's untouched front-end does)."""
from __future__ import annotations

import struct
from dataclasses import dataclass, field

import numpy as np


def _varint(buf: bytes, pos: int):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def fields(buf: bytes):
    """Yield (field number, wire type, value) of one message: value is an int (varint / fixed) or bytes (length-delimited)."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError(f"unsupported wire type {wt} (field {fno})")
        yield fno, wt, v


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_ints(wt: int, v) -> list[int]:
    if wt == 0:
        return [_signed64(v)]
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(_signed64(x))
    return out


def _packed_floats(wt: int, v) -> list[float]:
    if wt == 5:
        return [struct.unpack("<f", struct.pack("<I", v))[0]]
    return list(struct.unpack(f"<{len(v) // 4}f", v))


_NP = {1: np.float32, 2: np.uint8, 3: np.int8, 5: np.int16, 6: np.int32, 7: np.int64, 9: np.bool_, 10: np.float16, 11: np.float64}


@dataclass
class Tensor:
    name: str = ""
    dims: list = field(default_factory=list)
    data_type: int = 0
    raw: bytes = b""
    floats: list = field(default_factory=list)
    ints: list = field(default_factory=list)

    def numpy(self) -> np.ndarray:
        dt = _NP[self.data_type]
        if self.raw:
            a = np.frombuffer(self.raw, dtype=dt)
        elif self.floats:
            a = np.asarray(self.floats, dtype=dt)
        else:
            a = np.asarray(self.ints, dtype=dt)
        return a.reshape(self.dims) if self.dims else a.reshape(())


def parse_tensor(buf: bytes) -> Tensor:
    t = Tensor()
    for fno, wt, v in fields(buf):
        if fno == 1:
            t.dims.extend(_packed_ints(wt, v))
        elif fno == 2:
            t.data_type = v
        elif fno == 4:
            t.floats.extend(_packed_floats(wt, v))
        elif fno in (5, 7):
            t.ints.extend(_packed_ints(wt, v))
        elif fno == 8:
            t.name = v.decode()
        elif fno == 9:
            t.raw = v
    return t


@dataclass
class Node:
    op_type: str = ""
    name: str = ""
    inputs: list = field(default_factory=list)
    outputs: list = field(default_factory=list)
    attrs: dict = field(default_factory=dict)


def parse_node(buf: bytes) -> Node:
    nd = Node()
    for fno, wt, v in fields(buf):
        if fno == 1:
            nd.inputs.append(v.decode())
        elif fno == 2:
            nd.outputs.append(v.decode())
        elif fno == 3:
            nd.name = v.decode()
        elif fno == 4:
            nd.op_type = v.decode()
        elif fno == 5:
            name, val, ints, floats = "", None, [], []
            for f2, w2, v2 in fields(v):
                if f2 == 1:
                    name = v2.decode()
                elif f2 == 2:
                    val = struct.unpack("<f", struct.pack("<I", v2))[0]
                elif f2 == 3:
                    val = _signed64(v2)
                elif f2 == 4:
                    val = v2.decode(errors="replace")
                elif f2 == 5:
                    val = parse_tensor(v2)
                elif f2 == 7:
                    floats.extend(_packed_floats(w2, v2))
                elif f2 == 8:
                    ints.extend(_packed_ints(w2, v2))
            nd.attrs[name] = ints if ints else (floats if floats else val)
    return nd


@dataclass
class ValueInfo:
    name: str = ""
    elem_type: int = 0
    shape: list = field(default_factory=list)


def parse_value_info(buf: bytes) -> ValueInfo:
    vi = ValueInfo()
    for fno, _, v in fields(buf):
        if fno == 1:
            vi.name = v.decode()
        elif fno == 2:
            for f2, _, v2 in fields(v):
                if f2 == 1:  # tensor_type
                    for f3, _, v3 in fields(v2):
                        if f3 == 1:
                            vi.elem_type = v3
                        elif f3 == 2:
                            for f4, _, v4 in fields(v3):
                                if f4 == 1:
                                    dim = None
                                    for f5, _, v5 in fields(v4):
                                        dim = _signed64(v5) if f5 == 1 else v5.decode()
                                    vi.shape.append(dim)
    return vi


@dataclass
class Graph:
    nodes: list
    initializers: dict
    inputs: list
    outputs: list
    opset: int


def load(buf: bytes) -> Graph:
    graph, opset = None, 0
    for fno, _, v in fields(buf):
        if fno == 7:
            graph = v
        elif fno == 8:
            dom, ver = "", 0
            for f2, _, v2 in fields(v):
                if f2 == 1:
                    dom = v2.decode()
                elif f2 == 2:
                    ver = v2
            if dom in ("", "ai.onnx"):
                opset = ver
    if graph is None:
        raise ValueError("not a ModelProto: no graph field")
    nodes, inits, ins, outs = [], {}, [], []
    for fno, _, v in fields(graph):
        if fno == 1:
            nodes.append(parse_node(v))
        elif fno == 5:
            t = parse_tensor(v)
            inits[t.name] = t
        elif fno == 11:
            ins.append(parse_value_info(v))
        elif fno == 12:
            outs.append(parse_value_info(v))
    return Graph(nodes, inits, [i for i in ins if i.name not in inits], outs, opset)
