"""The `onnx.helper` constructors the front-end (onnx.py:10-16) and the tests use."""
from __future__ import annotations

import numbers

import numpy as np

from . import (IR_VERSION, AttributeProto, GraphProto, ModelProto, NodeProto, OperatorSetIdProto, TensorProto, ValueInfoProto)
from .numpy_helper import _NP

_DEFAULT_OPSET = 17


def make_attribute(key: str, value):
    a = AttributeProto()
    a.name = key
    if isinstance(value, (bool, np.bool_)):
        a.i, a.type = int(value), AttributeProto.INT
    elif isinstance(value, numbers.Integral):
        a.i, a.type = int(value), AttributeProto.INT
    elif isinstance(value, numbers.Real):
        a.f, a.type = float(value), AttributeProto.FLOAT
    elif isinstance(value, str):
        a.s, a.type = value.encode("utf-8"), AttributeProto.STRING
    elif isinstance(value, bytes):
        a.s, a.type = value, AttributeProto.STRING
    elif isinstance(value, TensorProto):
        a.t.CopyFrom(value)
        a.type = AttributeProto.TENSOR
    elif isinstance(value, GraphProto):
        a.g.CopyFrom(value)
        a.type = AttributeProto.GRAPH
    else:
        vals = list(value)
        if all(isinstance(v, numbers.Integral) for v in vals):
            a.ints.extend(int(v) for v in vals)
            a.type = AttributeProto.INTS
        elif all(isinstance(v, numbers.Real) for v in vals):
            a.floats.extend(float(v) for v in vals)
            a.type = AttributeProto.FLOATS
        elif all(isinstance(v, (str, bytes)) for v in vals):
            a.strings.extend(v.encode("utf-8") if isinstance(v, str) else v for v in vals)
            a.type = AttributeProto.STRINGS
        elif all(isinstance(v, TensorProto) for v in vals):
            a.tensors.extend(vals)
            a.type = AttributeProto.TENSORS
        else:
            raise ValueError(f"attribute {key}: unsupported value {value!r}")
    return a


def get_attribute_value(attr):
    T = AttributeProto
    return {T.FLOAT: lambda: attr.f, T.INT: lambda: attr.i, T.STRING: lambda: attr.s, T.TENSOR: lambda: attr.t, T.GRAPH: lambda: attr.g,
            T.FLOATS: lambda: list(attr.floats), T.INTS: lambda: list(attr.ints), T.STRINGS: lambda: list(attr.strings),
            T.TENSORS: lambda: list(attr.tensors), T.GRAPHS: lambda: list(attr.graphs)}[attr.type]()


def make_node(op_type, inputs, outputs, name=None, doc_string=None, domain=None, **kwargs):
    n = NodeProto()
    n.op_type = op_type
    n.input.extend(inputs)
    n.output.extend(outputs)
    if name:
        n.name = name
    if doc_string:
        n.doc_string = doc_string
    if domain is not None:
        n.domain = domain
    for k in sorted(kwargs):
        if kwargs[k] is not None:
            n.attribute.append(make_attribute(k, kwargs[k]))
    return n


def make_tensor(name, data_type, dims, vals, raw=False):
    t = TensorProto()
    t.name, t.data_type = name, data_type
    t.dims.extend(dims)
    if raw:
        t.raw_data = bytes(vals)
        return t
    flat = np.asarray(vals).ravel()
    if data_type == TensorProto.FLOAT:
        t.float_data.extend(float(v) for v in flat)
    elif data_type == TensorProto.DOUBLE:
        t.double_data.extend(float(v) for v in flat)
    elif data_type == TensorProto.INT64:
        t.int64_data.extend(int(v) for v in flat)
    elif data_type in (TensorProto.UINT32, TensorProto.UINT64):
        t.uint64_data.extend(int(v) for v in flat)
    elif data_type == TensorProto.FLOAT16:
        t.int32_data.extend(int(v) for v in np.asarray(flat, dtype=np.float16).view(np.uint16))
    elif data_type in _NP:
        t.int32_data.extend(int(v) for v in flat)
    else:
        raise ValueError(f"make_tensor: unsupported data type {data_type}")
    return t


def make_tensor_value_info(name, elem_type, shape, doc_string="", shape_denotation=None):
    v = ValueInfoProto()
    v.name = name
    if doc_string:
        v.doc_string = doc_string
    tt = v.type.tensor_type
    tt.elem_type = elem_type
    if shape is not None:
        tt.shape.SetInParent()
        for d in shape:
            dim = tt.shape.dim.add()
            if d is None:
                pass
            elif isinstance(d, str):
                dim.dim_param = d
            else:
                dim.dim_value = int(d)
    return v


def make_graph(nodes, name, inputs, outputs, initializer=None, doc_string=None, value_info=None, sparse_initializer=None):
    g = GraphProto()
    g.node.extend(nodes)
    g.name = name
    g.input.extend(inputs)
    g.output.extend(outputs)
    g.initializer.extend(initializer or [])
    g.value_info.extend(value_info or [])
    if doc_string:
        g.doc_string = doc_string
    return g


def make_opsetid(domain, version):
    o = OperatorSetIdProto()
    o.domain, o.version = domain, version
    return o


make_operatorsetid = make_opsetid


def make_model(graph, **kwargs):
    m = ModelProto()
    m.ir_version = kwargs.pop("ir_version", IR_VERSION)
    m.graph.CopyFrom(graph)
    opsets = kwargs.pop("opset_imports", None)
    if opsets:
        m.opset_import.extend(opsets)
    else:
        m.opset_import.add().version = _DEFAULT_OPSET
    for k, v in kwargs.items():
        setattr(m, k, v)
    return m


def tensor_dtype_to_np_dtype(t):
    return np.dtype(_NP[t])
