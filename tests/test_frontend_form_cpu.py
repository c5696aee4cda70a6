"""The front-end FORM, pinned to real exports (round-3 verdict "do this" #7; SURVEY §8 row g / f3).

tests/golden/onnx/*.onnx are the bytes `torch.onnx.export` wrote for an HF BertLayer (opset 13 and 17) and a small ResNet
(generator: tests/golden/make_onnx_fixtures.py). They are read with a wire-format reader (tests/onnx_wire.py — no `onnx`
package in the image), lowered exactly as pyinfinitensor/onnx.py lowers them (tests/onnx_import.py, a line-cited mirror of
OnnxStub.__init__) into a reference graph, and then:
  * the operator sequence must EQUAL what the hand-written builders of tools/model_bench.py (the graphs bench.py and the
    model tests run) emit for the same topology — so the builders are no longer a reading of onnx.py but a checked copy;
  * the launch plan of the imported graph must be the one the builders' graphs get (every chain fused).
Numerics of the same imported graphs against torch's own outputs: tests/test_gpu_frontend_exports.py (needs the GPU)."""
import sys
from collections import Counter
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import onnx_import as OI  # noqa: E402
import onnx_wire as W  # noqa: E402

GOLD = Path(__file__).resolve().parent / "golden" / "onnx"


@pytest.fixture(scope="module")
def B(plugin_backend):
    if not hasattr(plugin_backend.GraphHandler, "rocm_fusion_plan"):
        pytest.skip("plugin build predates the planner")
    return plugin_backend


def test_wire_reader_on_the_fixtures():
    g = W.load((GOLD / "bert_layer_tiny_opset13.onnx").read_bytes())
    assert g.opset == 13 and [(i.name, i.shape) for i in g.inputs] == [("x", [2, 16, 128]), ("mask", [2, 1, 1, 16])]
    hist = Counter(n.op_type for n in g.nodes)
    # a transformer layer as an opset-13 export: 8 MatMuls (q, k, v, QK^T, PV, o, ffn1, ffn2), LayerNorm and Gelu in primitives
    assert hist["MatMul"] == 8 and hist["Softmax"] == 1 and hist["Erf"] == 1 and hist["ReduceMean"] == 4 and hist["Pow"] == 2
    assert "LayerNormalization" not in hist and "Gelu" not in hist
    k_t = [n for n in g.nodes if n.op_type == "Transpose" and n.attrs["perm"] == [0, 2, 3, 1]]
    assert len(k_t) == 1, "K reaches Q.K^T through ONE Transpose(0, 2, 3, 1) in a transformers-5.x export"
    scale = [n for n in g.nodes if n.op_type == "Mul" and any(i.endswith("Constant_3_output_0") for i in n.inputs)]
    assert scale and abs(float(next(c for c in g.nodes if c.outputs == [scale[0].inputs[1]]).attrs["value"].numpy()) - 0.125) < 1e-6
    bias_first = [n for n in g.nodes if n.op_type == "Add" and n.inputs[0].endswith(".bias")]
    assert len(bias_first) == 6, "every nn.Linear is MatMul -> Add(bias, .) with the bias as the FIRST operand"
    g17 = W.load((GOLD / "bert_layer_tiny_opset17.onnx").read_bytes())
    h17 = Counter(n.op_type for n in g17.nodes)
    assert g17.opset == 17 and h17["LayerNormalization"] == 2 and h17["Erf"] == 1 and "ReduceMean" not in h17
    gr = W.load((GOLD / "resnet_tiny_opset13.onnx").read_bytes())
    hr = Counter(n.op_type for n in gr.nodes)
    assert hr == Counter({"Conv": 12, "Relu": 10, "Add": 3, "MaxPool": 1, "GlobalAveragePool": 1, "Flatten": 1, "Gemm": 1})
    assert all(len(n.inputs) == 3 for n in gr.nodes if n.op_type == "Conv"), "BatchNorm folded: every Conv carries a bias"
    assert "BatchNormalization" not in hr


def _imported(B, name, half):
    h, T, feeds, ins, outs = OI.import_graph(B, B.cpu_runtime(), (GOLD / name).read_bytes(), half=half)
    h.data_malloc()
    for t, a in feeds:  # the planner reads the scalar constants (Pow's 2, Gelu's sqrt 2 / 1 / 0.5, epsilon)
        t.copyin_numpy(np.ascontiguousarray(a))
    return h


def _strip(plan):
    """Plan items without positions: what is fused with what, in order."""
    out = []
    for line in plan:
        what, mem = line.split(" ", 1)[1].rsplit(" [", 1)
        out.append((what, len(mem.split(","))))
    return out


@pytest.mark.parametrize("opset,decomposed", [(13, True), (17, "gelu")])
def test_bert_builder_equals_the_real_export(B, opset, decomposed):
    """One encoder layer: tools/model_bench.py::build_bert(exporter="hf5") against the imported export — same operator
    sequence, same launch plan (f16: grouped q / k / v, fused attention, bias / Gelu epilogues, Add -> LayerNorm)."""
    from model_bench import Builder, build_bert

    for half in (False, True):
        hi = _imported(B, f"bert_layer_tiny_opset{opset}.onnx", half)
        bl = Builder(B, B.cpu_runtime(), "f16" if half else "f32", seed=0)
        build_bert(bl, 2, 16, 1, hidden=128, heads=2, ffn=256, vocab=50, decomposed=decomposed, exporter="hf5")
        bl.finish()
        want, got = OI.op_names(hi), OI.op_names(bl.h)
        first = got.index("MatMul")  # the builder starts with the embedding lookup + LayerNorm, the fixture is one BertLayer
        assert got[first:] == want, (got[first:], want)
        plan_i = _strip(hi.rocm_fusion_plan())
        plan_b = [p for p in _strip(bl.h.rocm_fusion_plan())]
        k = next(i for i, (w, _) in enumerate(plan_b) if w.startswith("matmul"))
        assert plan_b[k:] == plan_i, (plan_b[k:], plan_i)
        if half:
            whats = [w for w, _ in plan_i]
            assert whats[0].startswith("matmul+bias+headsplit x3 (grouped)") and whats[1] == "attention", whats
            assert any(w.startswith("matmul+bias+gelu") for w in whats) and sum("layernorm" in w for w in whats) == 2, whats
            assert not any(w == "op" for w in whats), whats  # nothing of the layer runs alone


def test_resnet_builder_equals_the_real_export(B):
    """Stem + three bottlenecks + pool + classifier: build_resnet50 on the fixture's topology against the imported export."""
    from model_bench import Builder, build_resnet50

    for half in (False, True):
        hi = _imported(B, "resnet_tiny_opset13.onnx", half)
        bl = Builder(B, B.cpu_runtime(), "f16" if half else "f32", seed=0)
        build_resnet50(bl, 2, image=32, stem=16, stages=((8, 2, 1), (16, 1, 2)), classes=10)
        bl.finish()
        assert OI.op_names(bl.h) == OI.op_names(hi)
        assert _strip(bl.h.rocm_fusion_plan()) == _strip(hi.rocm_fusion_plan())
        whats = [w for w, _ in _strip(hi.rocm_fusion_plan())]
        assert sum(w.startswith("conv+bias") for w in whats) == 12, whats  # every conv carries its (Reshape'd) bias
