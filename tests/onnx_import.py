"""A line-by-line mirror of the part of `OnnxStub.__init__` (pyinfinitensor/src/pyinfinitensor/onnx.py:41-1136) that the
exports under tests/golden/onnx/ exercise: ONNX nodes -> `backend.GraphHandler` calls in the reference's own lowering —
Conv with bias -> conv, reshape(bias, [1, F, 1, 1]), add (:159-190); MatMul without bias or transposes (:280-290); Gemm
with its bias and transB (:291-311); MaxPool (:374-425); GlobalAveragePool as avgPool over the plane (:489-503); the
element-wise operators; Softmax (:590-598); Identity (:619-623); Flatten (:624-632); Transpose (:646-655); Reshape with a
static shape input (:671-677); ReduceMean with `axes` as attribute or input (:837-855); Erf; Constant -> a weight tensor
(:1088-1095); LayerNormalization. Initializers become weights, graph inputs become inputs, graph outputs are marked (:1118).

The reference first runs `onnxsim.simplify` (:49-57); neither `onnx` nor `onnxsim` exists in this image, so the RAW export is
imported — a superset of what the front-end sees (simplification only folds Constant nodes into initializers and drops
Identity copies). Test infrastructure: the product never parses ONNX."""
from __future__ import annotations

import numpy as np

import onnx_wire as W

F32, F16 = 1, 10


def import_graph(B, runtime, model_bytes: bytes, half: bool = False):
    """-> (handler, {name: tensor}, [(tensor, array)] weights to copy in, [input names], [output names]).
    half: build the graph in f16 (float tensors and constants converted) — the dtype of BASELINE configs 3 / 4."""
    g = W.load(model_bytes)
    h = B.GraphHandler(runtime)
    tensors, data, feeds = {}, {}, []
    fcode = F16 if half else F32

    def code_of(onnx_dt):
        return fcode if onnx_dt in (1, 10, 11) else onnx_dt

    def as_np(t: W.Tensor):
        a = t.numpy()
        return a.astype(np.float16 if half else np.float32) if t.data_type in (1, 10, 11) else a

    for name, t in g.initializers.items():  # :139-143
        tensors[name] = h.tensor(list(t.dims), code_of(t.data_type))
        tensors[name].set_weight()
        data[name] = t
        feeds.append((tensors[name], as_np(t).reshape(t.dims)))
    for vi in g.inputs:  # :145-151
        tensors[vi.name] = h.tensor(list(vi.shape), code_of(vi.elem_type))
        tensors[vi.name].set_input()

    def static(node, idx):  # _parse_static_input (:1545-1567)
        name = node.inputs[idx]
        if name not in data:
            raise ValueError(f'{node.op_type} input {idx} ("{name}") must be constant')
        return [int(v) if float(v).is_integer() else float(v) for v in data[name].numpy().ravel().tolist()]

    T = tensors
    lin = B.ActType.Linear
    for nd in g.nodes:  # the exporter emits nodes in topological order (the reference sorts them first: :82-122)
        i, o, a, op = nd.inputs, nd.outputs, nd.attrs, nd.op_type
        if op == "Conv":
            d, p, s = a.get("dilations", [1, 1]), a.get("pads", [0, 0, 0, 0]), a.get("strides", [1, 1])
            assert p[0] == p[2] and p[1] == p[3] and a.get("group", 1) == 1
            if len(i) > 2:
                conv = h.conv(T[i[0]], T[i[1]], None, p[0], p[1], s[0], s[1], d[0], d[1])
                n_b = int(np.prod(T[i[2]].shape()))
                T[o[0]] = h.add(conv, h.reshape(T[i[2]], None, [1, n_b, 1, 1]), None)
            else:
                T[o[0]] = h.conv(T[i[0]], T[i[1]], None, p[0], p[1], s[0], s[1], d[0], d[1])
        elif op == "MatMul":
            T[o[0]] = h.matmul(T[i[0]], T[i[1]], None, False, False, None, lin, "default")
        elif op == "Gemm":
            assert a.get("alpha", 1.0) == 1.0 and a.get("beta", 1.0) == 1.0
            T[o[0]] = h.matmul(T[i[0]], T[i[1]], None, a.get("transA", 0) == 1, a.get("transB", 0) == 1, T[i[2]] if len(i) > 2 else None,
                               lin, "default")
        elif op == "MaxPool":
            k, d, p, s = a["kernel_shape"], a.get("dilations", [1, 1]), a.get("pads", [0, 0, 0, 0]), a.get("strides", [1, 1])
            assert p[0] == p[2] and p[1] == p[3]
            T[o[0]] = h.maxPool(T[i[0]], None, k[0], k[1], d[0], d[1], p[0], p[1], s[0], s[1], a.get("ceil_mode", 0))
        elif op == "GlobalAveragePool":
            _, _, hh, ww = T[i[0]].shape()
            T[o[0]] = h.avgPool(T[i[0]], None, hh, ww, 1, 1, 0, 0, 1, 1, 0)
        elif op in ("Add", "Sub", "Mul", "Div", "Pow"):
            T[o[0]] = getattr(h, op.lower())(T[i[0]], T[i[1]], None)
        elif op in ("Relu", "Sqrt", "Erf", "Identity", "Sigmoid", "Tanh"):
            T[o[0]] = getattr(h, op.lower())(T[i[0]], None)
        elif op == "Softmax":
            T[o[0]] = h.softmax(T[i[0]], None, a.get("axis", -1))
        elif op == "Flatten":
            T[o[0]] = h.flatten(T[i[0]], None, a.get("axis", 1))
        elif op == "Transpose":
            T[o[0]] = h.transpose(T[i[0]], None, a["perm"])
        elif op == "Reshape":
            T[o[0]] = h.reshape(T[i[0]], None, static(nd, 1))
        elif op == "ReduceMean":
            axes = a.get("axes") or (static(nd, 1) if len(i) > 1 and i[1] else None)
            T[o[0]] = h.reduceMean(T[i[0]], None, axes, a.get("keepdims", 1) != 0)
        elif op == "LayerNormalization":
            T[o[0]] = h.layerNormalization(T[i[0]], T[i[1]], None, T[i[2]] if len(i) > 2 else None, a.get("epsilon", 1e-5),
                                           a.get("axis", -1), a.get("stash_type", 1))
        elif op == "Constant":
            t = a["value"]
            T[o[0]] = h.tensor(list(t.dims), code_of(t.data_type))
            T[o[0]].set_weight()
            data[o[0]] = t
            feeds.append((T[o[0]], as_np(t).reshape(t.dims)))
        else:
            raise NotImplementedError(f'operator "{op}" is not in this mirror of onnx.py')
    for vo in g.outputs:
        T[vo.name].set_output()
    return h, T, feeds, [vi.name for vi in g.inputs], [vo.name for vo in g.outputs]


def op_names(h) -> list[str]:
    return [str(o.op_type().id()).split(".")[-1] for o in h.operators()]
