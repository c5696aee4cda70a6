"""OnnxStub smoke path: the reference's ONNX front-end (pyinfinitensor/src/pyinfinitensor/onnx.py:41)
driving Device::ROCM unchanged — `OnnxStub(model, backend.RocmRuntime(0)).run()` against the same model on
`backend.cpu_runtime()`.

The front-end needs `onnx` + `onnxsim` (absent from this image: tests/onnx_shim supplies protobuf-backed stand-ins) and the reference's
own Python file (never copied into this repo: found through an installed `pyinfinitensor`, $INFINITENSOR_PY_SRC, or
/root/reference/pyinfinitensor/src where the reference checkout exists — the GPU boxes have none, so THERE this test skips and
tests/test_gpu_frontend_exports.py::test_graph_on_rocm_is_the_real_front_ends_graph carries the claim; the same OnnxStub runs in the CPU
suite of the build container, tests/test_frontend_real_cpu.py)."""
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _onnx_stub(plugin_backend):
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import frontend_real as FR

    mod = FR.load_frontend(plugin_backend)  # (real `onnx` / `onnxsim` when installed, else the protobuf stand-ins of tests/onnx_shim)
    if mod is None:
        pytest.skip("the reference's pyinfinitensor/onnx.py is not on this machine: it is never shipped with this repository (set "
                    "INFINITENSOR_PY_SRC=<reference>/pyinfinitensor/src); tests/test_gpu_frontend_exports.py proves instead that the graphs "
                    "run here carry the real front-end's signature")
    return mod


def _model():
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import frontend_real as FR

    FR.ensure_onnx()
    import onnx
    from onnx import TensorProto, helper, numpy_helper

    rng = np.random.default_rng(0)
    w = rng.standard_normal((16, 8)).astype(np.float32)
    b = rng.standard_normal((8,)).astype(np.float32)
    nodes = [helper.make_node("MatMul", ["x", "w"], ["mm"]), helper.make_node("Add", ["mm", "b"], ["s"]),
             helper.make_node("Relu", ["s"], ["y"])]
    graph = helper.make_graph(nodes, "mlp", [helper.make_tensor_value_info("x", TensorProto.FLOAT, [4, 16])],
                              [helper.make_tensor_value_info("y", TensorProto.FLOAT, [4, 8])],
                              [numpy_helper.from_array(w, "w"), numpy_helper.from_array(b, "b")])
    model = helper.make_model(graph, opset_imports=[helper.make_opsetid("", 13)])
    onnx.checker.check_model(model)
    return model, w, b


def test_onnx_stub_runs_on_rocm_and_matches_cpu(plugin_backend):
    front = _onnx_stub(plugin_backend)
    B = plugin_backend
    model, w, b = _model()
    x = np.random.default_rng(1).standard_normal((4, 16)).astype(np.float32)
    outs = {}
    for name, rt in (("rocm", B.RocmRuntime(0)), ("cpu", B.cpu_runtime())):
        stub = front.OnnxStub(model, rt)
        next(iter(stub.inputs.values())).copyin_numpy(x)
        stub.run()
        outs[name] = next(iter(stub.outputs.values())).copyout_numpy()
        if name == "rocm":  # the hipGraph path: what the one-line `run_with_hipgraph` of INTEGRATION.md section 2 calls
            stub.handler.run_with_hipgraph()
            assert np.array_equal(next(iter(stub.outputs.values())).copyout_numpy(), outs[name])
    want = np.maximum(x.astype(np.float64) @ w + b, 0)
    assert np.allclose(outs["rocm"], want, rtol=1e-4, atol=1e-5)
    assert np.allclose(outs["rocm"], outs["cpu"], rtol=1e-4, atol=1e-5)
