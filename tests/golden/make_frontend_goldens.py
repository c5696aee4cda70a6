#!/usr/bin/env python3
"""Golden GRAPH SIGNATURES and OUTPUTS of the reference's real front-end on the committed ONNX exports.

Runs where /root/reference exists (the build container): loads the reference's unmodified `OnnxStub`
(pyinfinitensor/src/pyinfinitensor/onnx.py:41-1136, via tests/frontend_real.py — protobuf-backed `onnx` stand-in when the real package
is absent), imports every tests/golden/onnx/*.onnx on the reference's own `backend.cpu_runtime()`, and writes
  <model>_frontend.json      the signature (tests/graph_signature.py) of the graph OnnxStub built — DATA: op list, attributes, shapes
  <model>_frontend_out.npz   that graph's outputs on the fixture's inputs, computed by the reference's native-CPU kernels
                             (only when every operator has a native-CPU kernel; `ran` in the json says so)
The GPU boxes have no /root/reference: there tests/test_gpu_frontend_exports.py checks that the mirror importer builds a graph with the SAME
signature on Device::ROCM and that its outputs match these. tests/test_frontend_real_cpu.py re-derives the signatures wherever the
reference is present, so a stale golden fails the CPU suite."""
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
TESTS = HERE.parent
sys.path.insert(0, str(TESTS))
sys.path.insert(0, str(TESTS.parent))
import frontend_real as FR  # noqa: E402
import graph_signature as GS  # noqa: E402
from conftest import load_backend_module  # noqa: E402

MODELS = {"resnet_tiny_opset13": "resnet_tiny_io.npz", "bert_layer_tiny_opset13": "bert_layer_tiny_io.npz", "bert_layer_tiny_opset17": "bert_layer_tiny_io.npz"}


def build(B, front, name):
    import onnx

    model = onnx.load(str(HERE / "onnx" / f"{name}.onnx"))
    stub = front.OnnxStub(model, B.cpu_runtime())
    return stub, GS.signature(B, stub.handler)


def main():
    B = load_backend_module()
    front = FR.load_frontend(B)
    assert front is not None, "the reference's pyinfinitensor/onnx.py is not available here"
    for name, io_name in MODELS.items():
        stub, sig = build(B, front, name)
        io = np.load(HERE / "onnx" / io_name)
        rec = {"model": f"{name}.onnx", "frontend": "pyinfinitensor/src/pyinfinitensor/onnx.py (OnnxStub, unmodified)", "onnx_package": FR.ensure_onnx(),
               "inputs": list(stub.inputs.keys()), "outputs": list(stub.outputs.keys()), "signature": sig}
        try:
            for k, t in stub.inputs.items():
                t.copyin_numpy(np.ascontiguousarray(io[k]))
            stub.run()
            outs = {k: t.copyout_numpy() for k, t in stub.outputs.items()}
            np.savez_compressed(HERE / "onnx" / f"{name}_frontend_out.npz", **outs)
            rec["ran"] = True
            rec["max_abs_diff_vs_torch"] = {k: float(np.abs(v.astype(np.float64).reshape(io[k].shape) - io[k]).max()) for k, v in outs.items() if k in io}
        except Exception as e:  # an operator without a native-CPU kernel: the signature is still the front-end's
            rec["ran"] = False
            rec["why_not"] = repr(e)[:300]
        (HERE / "onnx" / f"{name}_frontend.json").write_text(json.dumps(rec, indent=1) + "\n")
        print(name, len(sig["ops"]), "operators; ran on cpu_runtime:", rec["ran"], rec.get("max_abs_diff_vs_torch", rec.get("why_not")))


if __name__ == "__main__":
    main()
