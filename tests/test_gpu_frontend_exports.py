"""REAL exports through the plugin (the OnnxStub path itself needs the `onnx` package, which the image lacks):
the ModelProto bytes under tests/golden/onnx/ (torch.onnx.export of an HF BertLayer and a small ResNet), lowered like
pyinfinitensor/onnx.py lowers them (tests/onnx_import.py), run on Device::ROCM by the reference executor — planned launches,
eager and hipGraph — against TORCH'S OWN fp32 outputs stored with the fixtures: fp32 within 1e-4 of the output scale
(`north_star`), f16 at storage tolerance. This is the only test whose expected values come from outside this repository's
oracle AND whose graph comes from outside its builders."""
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent))
import json  # noqa: E402

import graph_signature as GS  # noqa: E402
import onnx_import as OI  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden" / "onnx"


@pytest.mark.parametrize("half,tol", [(False, 1e-4), (True, 2e-2)])
@pytest.mark.parametrize("model,io", [("bert_layer_tiny_opset13", "bert_layer_tiny_io"), ("bert_layer_tiny_opset17", "bert_layer_tiny_io"),
                                      ("resnet_tiny_opset13", "resnet_tiny_io")])
def test_exported_graph_on_rocm_vs_torch(plugin_backend, model, io, half, tol):
    B = plugin_backend
    rocm = B.RocmRuntime(0)
    data = np.load(GOLD / f"{io}.npz")
    want = data["y"].astype(np.float64)
    scale = np.abs(want).max()
    res = {}
    try:
        for mode in ("fused", "unfused", "hipgraph"):
            rocm.set_fusion(mode != "unfused")
            h, T, feeds, ins, outs = OI.import_graph(B, rocm, (GOLD / f"{model}.onnx").read_bytes(), half=half)
            h.data_malloc()
            for t, a in feeds:
                t.copyin_numpy(np.ascontiguousarray(a))
            for n in ins:
                T[n].copyin_numpy(np.ascontiguousarray(data[n].astype(np.float16 if half else np.float32)))
            if mode == "hipgraph":
                h.run_with_hipgraph()
                h.run_with_hipgraph()
            else:
                f0 = rocm.fused_launch_count()
                h.run()
                if mode == "fused":
                    plan = h.rocm_fusion_plan()
                    assert rocm.fused_launch_count() - f0 >= (6 if "bert" in model else 12), plan
            res[mode] = T[outs[0]].copyout_numpy().astype(np.float64).reshape(want.shape)
    finally:
        rocm.set_fusion(True)
    for mode, got in res.items():
        assert np.isfinite(got).all(), mode
        assert np.abs(got - want).max() <= tol * scale, (mode, np.abs(got - want).max(), scale)
    assert np.array_equal(res["fused"], res["hipgraph"])


@pytest.mark.parametrize("model", ["bert_layer_tiny_opset13", "bert_layer_tiny_opset17", "resnet_tiny_opset13"])
def test_graph_on_rocm_is_the_real_front_ends_graph(plugin_backend, model):
    """The graph the tests above run on Device::ROCM has the SIGNATURE (operators, attributes, edges, shapes, dtypes) of the graph the
    reference's unmodified OnnxStub builds from the same bytes — the golden was written by running the real front-end in the build
    container (tests/golden/make_frontend_goldens.py) and is re-derived there every round (tests/test_frontend_real_cpu.py)."""
    B = plugin_backend
    gold = json.loads((GOLD / f"{model}_frontend.json").read_text())
    assert gold["frontend"].endswith("(OnnxStub, unmodified)")
    h, T, feeds, ins, outs = OI.import_graph(B, B.RocmRuntime(0), (GOLD / f"{model}.onnx").read_bytes(), half=False)
    assert GS.diff(GS.signature(B, h), gold["signature"]) is None
    assert ins == gold["inputs"] and outs == gold["outputs"]
