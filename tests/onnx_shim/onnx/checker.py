"""`onnx.checker` in the small: the structural checks that need no operator schemas (the real checker also validates every node against
the operator set; the front-end only calls these on what it built itself, onnx.py:1165-1215, and on nothing it imports)."""
from __future__ import annotations


class ValidationError(Exception):
    pass


def check_tensor(t, ctx=None):
    if t.data_type == 0:
        raise ValidationError(f"tensor {t.name!r}: undefined data type")
    n = 1
    for d in t.dims:
        if d < 0:
            raise ValidationError(f"tensor {t.name!r}: negative dimension")
        n *= d


def check_value_info(v, ctx=None):
    if not v.name:
        raise ValidationError("value info without a name")
    if not v.type.HasField("tensor_type") or v.type.tensor_type.elem_type == 0:
        raise ValidationError(f"value info {v.name!r}: no tensor element type")


def check_node(n, ctx=None):
    if not n.op_type:
        raise ValidationError("node without an op_type")
    if not n.output:
        raise ValidationError(f"node {n.name!r} ({n.op_type}) has no output")
    names = [a.name for a in n.attribute]
    if len(set(names)) != len(names):
        raise ValidationError(f"node {n.name!r}: duplicate attribute")


def check_graph(g, ctx=None):
    if not g.name:
        raise ValidationError("graph without a name")
    known = {i.name for i in g.input} | {t.name for t in g.initializer}
    for t in g.initializer:
        check_tensor(t)
    for v in list(g.input) + list(g.output):
        check_value_info(v)
    for n in g.node:  # nodes must be topologically sorted (IR specification)
        check_node(n)
        for i in n.input:
            if i and i not in known:
                raise ValidationError(f"node {n.name!r} ({n.op_type}): input {i!r} is not produced by an earlier node")
        known.update(o for o in n.output if o)
    for v in g.output:
        if v.name not in known:
            raise ValidationError(f"graph output {v.name!r} is not produced")


def check_model(m, full_check=False):
    if m.ir_version == 0:
        raise ValidationError("model without ir_version")
    if not m.opset_import:
        raise ValidationError("model without opset_import")
    check_graph(m.graph)
