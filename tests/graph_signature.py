"""A canonical, JSON-able description of the graph a `backend.GraphHandler` holds — what the front-end BUILT, independent of tensor
object identities: operators in the handler's order, each with its type, its attributes (through the reference's own `*_attrs_of` /
`*_of` accessors, ffi_infinitensor.cc:196-360) and its inputs / outputs as indices into a tensor table numbered by first appearance;
tensors as (shape, dtype). Two front-ends that lowered the same model the same way produce EQUAL signatures."""
from __future__ import annotations


def _op_name(op) -> str:
    return str(op.op_type().id()).split(".")[-1]


_ATTR = {
    "Conv": "conv_attrs_of", "ConvTranspose": "conv_trans_attrs_of", "MatMul": "matmul_attrs_of", "BatchNormalization": "batch_norm_attrs_of",
    "MaxPool": "pool_attrs_of", "AveragePool": "pool_attrs_of", "Clip": "clip_attrs_of", "ReduceMean": "reduce_attrs_of",
    "ReduceSum": "reduce_attrs_of", "Reshape": "reshape_shape_of", "Expand": "expand_shape_of", "Pad": "pad_pads_of",
    "Transpose": "transpose_permute_of", "Concat": "concat_axis_of", "Split": "split_axis_of", "Gather": "gather_axis_of",
    "Flatten": "flatten_axis_of", "Softmax": "softmax_axis_of", "Cast": "cast_to_of", "DepthToSpace": "depth_to_space_attrs_of",
    "Squeeze": "squeeze_axes_of", "Unsqueeze": "unsqueeze_axes_of", "LRN": "lrn_attrs_of", "Elu": "elu_alpha_of",
}


def _plain(v):
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if isinstance(v, bool):
        return bool(v)
    if isinstance(v, int):
        return int(v)
    if isinstance(v, float):
        return round(float(v), 7)
    return str(v)


def signature(B, handler) -> dict:
    table, index = [], {}

    def tid(t):
        k = t.fuid()
        if k not in index:
            index[k] = len(table)
            table.append({"shape": [int(d) for d in t.shape()], "dtype": int(B.tensor_dtype(t)) if hasattr(B, "tensor_dtype") else str(t.dtype())})
        return index[k]

    ops = []
    for op in handler.operators():
        name = _op_name(op)
        rec = {"op": name, "in": [tid(t) for t in op.inputs()], "out": [tid(t) for t in op.outputs()]}
        fn = _ATTR.get(name)
        if fn and hasattr(B, fn):
            rec["attrs"] = _plain(getattr(B, fn)(op))
        ops.append(rec)
    return {"ops": ops, "tensors": table}


def diff(a: dict, b: dict) -> str | None:
    """None when equal, else the first difference in words."""
    if len(a["ops"]) != len(b["ops"]):
        return f'{len(a["ops"])} operators vs {len(b["ops"])}: {[o["op"] for o in a["ops"]]} vs {[o["op"] for o in b["ops"]]}'
    for n, (x, y) in enumerate(zip(a["ops"], b["ops"])):
        if x != y:
            return f"operator {n}: {x} vs {y}"
    if a["tensors"] != b["tensors"]:
        for n, (x, y) in enumerate(zip(a["tensors"], b["tensors"])):
            if x != y:
                return f"tensor {n}: {x} vs {y}"
        return "tensor tables differ in length"
    return None
