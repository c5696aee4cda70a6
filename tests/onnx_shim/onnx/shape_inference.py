"""`onnx.shape_inference.infer_shapes`: imported by the front-end (onnx.py:25) but its only call site is commented out (:66-70) — the
reference infers shapes itself, operator by operator, in C++. Identity."""


def infer_shapes(model, check_type=False, strict_mode=False, data_prop=False):
    return model
