"""A stand-in for the `onnx` Python package (absent from this image; no network), for ONE purpose: letting the reference's UNMODIFIED
front-end (`pyinfinitensor/src/pyinfinitensor/onnx.py`, `OnnxStub`) import real `.onnx` files in the test suite.

The messages are REAL protobuf classes (google.protobuf is in the image), generated at import time from a schema written out below from
the public ONNX IR specification (onnx.proto: message names, field names, numbers and types — an interchange format, not code of the
reference). They therefore parse and serialise the same bytes the real package does and behave the same under `copy.deepcopy`,
`HasField`, repeated-field access, `TensorProto.FLOAT`-style enum constants, ...  `helper`, `numpy_helper`, `checker` and
`shape_inference` carry the functions the front-end calls. Put on sys.path (tests/frontend_real.py) ONLY when the real package cannot be
imported. Test infrastructure: the product never parses ONNX."""
from __future__ import annotations

from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_F = descriptor_pb2.FieldDescriptorProto
_OPT, _REP = _F.LABEL_OPTIONAL, _F.LABEL_REPEATED
_T = {"string": _F.TYPE_STRING, "bytes": _F.TYPE_BYTES, "int64": _F.TYPE_INT64, "int32": _F.TYPE_INT32, "uint64": _F.TYPE_UINT64,
      "float": _F.TYPE_FLOAT, "double": _F.TYPE_DOUBLE}


def _field(msg, name, number, typ, rep=False, oneof=None, packed=False):
    f = msg.field.add()
    f.name, f.number, f.label = name, number, _REP if rep else _OPT
    if typ in _T:
        f.type = _T[typ]
        if packed:  # (proto2: only the fields onnx.proto marks [packed = true] — the TensorProto payload arrays)
            f.options.packed = True
    elif typ.startswith("enum:"):
        f.type, f.type_name = _F.TYPE_ENUM, ".onnx." + typ[5:]
    else:
        f.type, f.type_name = _F.TYPE_MESSAGE, ".onnx." + typ
    if oneof is not None:
        f.oneof_index = oneof
    return f


def _enum(parent, name, values):
    e = parent.enum_type.add()
    e.name = name
    for k, v in values:
        ev = e.value.add()
        ev.name, ev.number = k, v


def _schema() -> descriptor_pb2.FileDescriptorProto:
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name, fd.package, fd.syntax = "onnx/onnx_shim.proto", "onnx", "proto2"

    m = fd.message_type.add(); m.name = "AttributeProto"
    _enum(m, "AttributeType", [("UNDEFINED", 0), ("FLOAT", 1), ("INT", 2), ("STRING", 3), ("TENSOR", 4), ("GRAPH", 5), ("FLOATS", 6),
                               ("INTS", 7), ("STRINGS", 8), ("TENSORS", 9), ("GRAPHS", 10)])
    for args in (("name", 1, "string"), ("ref_attr_name", 21, "string"), ("doc_string", 13, "string"),
                 ("type", 20, "enum:AttributeProto.AttributeType"), ("f", 2, "float"), ("i", 3, "int64"), ("s", 4, "bytes"),
                 ("t", 5, "TensorProto"), ("g", 6, "GraphProto")):
        _field(m, *args)
    for args in (("floats", 7, "float"), ("ints", 8, "int64"), ("strings", 9, "bytes"), ("tensors", 10, "TensorProto"),
                 ("graphs", 11, "GraphProto")):
        _field(m, *args, rep=True)

    m = fd.message_type.add(); m.name = "ValueInfoProto"
    _field(m, "name", 1, "string"); _field(m, "type", 2, "TypeProto"); _field(m, "doc_string", 3, "string")

    m = fd.message_type.add(); m.name = "NodeProto"
    _field(m, "input", 1, "string", rep=True); _field(m, "output", 2, "string", rep=True)
    _field(m, "name", 3, "string"); _field(m, "op_type", 4, "string"); _field(m, "domain", 7, "string")
    _field(m, "attribute", 5, "AttributeProto", rep=True); _field(m, "doc_string", 6, "string")

    m = fd.message_type.add(); m.name = "StringStringEntryProto"
    _field(m, "key", 1, "string"); _field(m, "value", 2, "string")

    m = fd.message_type.add(); m.name = "OperatorSetIdProto"
    _field(m, "domain", 1, "string"); _field(m, "version", 2, "int64")

    m = fd.message_type.add(); m.name = "ModelProto"
    _field(m, "ir_version", 1, "int64"); _field(m, "opset_import", 8, "OperatorSetIdProto", rep=True)
    _field(m, "producer_name", 2, "string"); _field(m, "producer_version", 3, "string"); _field(m, "domain", 4, "string")
    _field(m, "model_version", 5, "int64"); _field(m, "doc_string", 6, "string"); _field(m, "graph", 7, "GraphProto")
    _field(m, "metadata_props", 14, "StringStringEntryProto", rep=True)

    m = fd.message_type.add(); m.name = "GraphProto"
    _field(m, "node", 1, "NodeProto", rep=True); _field(m, "name", 2, "string")
    _field(m, "initializer", 5, "TensorProto", rep=True); _field(m, "doc_string", 10, "string")
    _field(m, "input", 11, "ValueInfoProto", rep=True); _field(m, "output", 12, "ValueInfoProto", rep=True)
    _field(m, "value_info", 13, "ValueInfoProto", rep=True)

    m = fd.message_type.add(); m.name = "TensorProto"
    _enum(m, "DataType", [("UNDEFINED", 0), ("FLOAT", 1), ("UINT8", 2), ("INT8", 3), ("UINT16", 4), ("INT16", 5), ("INT32", 6),
                          ("INT64", 7), ("STRING", 8), ("BOOL", 9), ("FLOAT16", 10), ("DOUBLE", 11), ("UINT32", 12), ("UINT64", 13),
                          ("COMPLEX64", 14), ("COMPLEX128", 15), ("BFLOAT16", 16)])
    _enum(m, "DataLocation", [("DEFAULT", 0), ("EXTERNAL", 1)])
    seg = m.nested_type.add(); seg.name = "Segment"
    _field(seg, "begin", 1, "int64"); _field(seg, "end", 2, "int64")
    _field(m, "dims", 1, "int64", rep=True); _field(m, "data_type", 2, "int32"); _field(m, "segment", 3, "TensorProto.Segment")
    _field(m, "float_data", 4, "float", rep=True, packed=True); _field(m, "int32_data", 5, "int32", rep=True, packed=True)
    _field(m, "string_data", 6, "bytes", rep=True); _field(m, "int64_data", 7, "int64", rep=True, packed=True)
    _field(m, "name", 8, "string"); _field(m, "doc_string", 12, "string"); _field(m, "raw_data", 9, "bytes")
    _field(m, "external_data", 13, "StringStringEntryProto", rep=True)
    _field(m, "data_location", 14, "enum:TensorProto.DataLocation")
    _field(m, "double_data", 10, "double", rep=True, packed=True); _field(m, "uint64_data", 11, "uint64", rep=True, packed=True)

    m = fd.message_type.add(); m.name = "TensorShapeProto"
    dim = m.nested_type.add(); dim.name = "Dimension"
    dim.oneof_decl.add().name = "value"
    _field(dim, "dim_value", 1, "int64", oneof=0); _field(dim, "dim_param", 2, "string", oneof=0); _field(dim, "denotation", 3, "string")
    _field(m, "dim", 1, "TensorShapeProto.Dimension", rep=True)

    m = fd.message_type.add(); m.name = "TypeProto"
    ten = m.nested_type.add(); ten.name = "Tensor"
    _field(ten, "elem_type", 1, "int32"); _field(ten, "shape", 2, "TensorShapeProto")
    m.oneof_decl.add().name = "value"
    _field(m, "tensor_type", 1, "TypeProto.Tensor", oneof=0); _field(m, "denotation", 6, "string")
    return fd


_pool = descriptor_pool.DescriptorPool()
_pool.Add(_schema())


def _cls(name):
    return message_factory.GetMessageClass(_pool.FindMessageTypeByName("onnx." + name))


AttributeProto = _cls("AttributeProto")
ValueInfoProto = _cls("ValueInfoProto")
NodeProto = _cls("NodeProto")
ModelProto = _cls("ModelProto")
GraphProto = _cls("GraphProto")
TensorProto = _cls("TensorProto")
TensorShapeProto = _cls("TensorShapeProto")
TypeProto = _cls("TypeProto")
OperatorSetIdProto = _cls("OperatorSetIdProto")
StringStringEntryProto = _cls("StringStringEntryProto")

IR_VERSION = 8
__version__ = "0.0-shim"
IS_SHIM = True


def load_model_from_string(data: bytes) -> "ModelProto":
    m = ModelProto()
    m.ParseFromString(data)
    return m


def load(path) -> "ModelProto":
    with open(path, "rb") as f:
        return load_model_from_string(f.read())


load_model = load


def save(model, path) -> None:
    with open(path, "wb") as f:
        f.write(model.SerializeToString())


from . import checker, helper, numpy_helper, shape_inference  # noqa: E402,F401
