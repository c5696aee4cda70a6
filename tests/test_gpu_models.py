"""End-to-end parity of a whole graph: the full ResNet-50 topology (53 convs, folded-BN bias
adds, ReLUs, residual joins, pools, classifier) built once per runtime with the REFERENCE's GraphHandler and run
(a) on Device::ROCM through the plugin — fp32 and fp16, with and without launch-time fusion, eager and hipGraph —
and (b) on the reference's own native-CPU runtime in fp32 in the same process. Same weights, same input (seeded)."""
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))

pytestmark = pytest.mark.gpu


def _logits(B, runtime, dtype, batch=1, image=32, hipgraph=False):
    from model_bench import Builder, build_resnet50

    bl = Builder(B, runtime, dtype, seed=0)
    out = build_resnet50(bl, batch, image, fc_bias_as_add=True)
    bl.finish()
    if hipgraph:
        bl.h.run_with_hipgraph()
        for t, a in bl.feeds:  # the planner may have recycled input storage: feed again before the replay
            t.copyin_numpy(np.ascontiguousarray(a))
        bl.h.run_with_hipgraph()
    else:
        bl.h.run()
    return out.copyout_numpy().astype(np.float64).reshape(batch, 1000)


def test_resnet50_logits_match_the_reference_cpu_backend(plugin_backend):
    B = plugin_backend
    want = _logits(B, B.cpu_runtime(), "f32")  # the reference's own kernels (naive conv, fp32)
    scale = np.abs(want).max()
    rocm = B.RocmRuntime(0)
    try:
        for fusion in (True, False):
            rocm.set_fusion(fusion)
            got32 = _logits(B, rocm, "f32")
            # fp32 gate: 1e-4 relative per operator (north_star); 53 convolutions deep the logits agree to ~1e-5 of scale
            assert np.abs(got32 - want).max() <= 1e-4 * scale, (fusion, np.abs(got32 - want).max(), scale)
            got16 = _logits(B, rocm, "f16")
            assert np.abs(got16 - want).max() <= 2e-2 * scale, (fusion, np.abs(got16 - want).max(), scale)
            assert (got16.argmax(1) == want.argmax(1)).all()
        rocm.set_fusion(True)
        gotg = _logits(B, rocm, "f32", hipgraph=True)
        assert np.abs(gotg - want).max() <= 1e-4 * scale
    finally:
        rocm.set_fusion(True)
