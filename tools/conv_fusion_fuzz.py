"""Random ResNet-style blocks through the reference executor + ROCM plugin, launch-time fusion ON vs OFF (f16): conv [+ bias]
[+ residual] [+ relu] chains, bottlenecks with and without a down-sampling branch, Relu -> MaxPool, global average pool. The
fused conv rounds once where the chain rounds per operator, so outputs are compared with a 16-bit tolerance; the point is to
catch a fused launch that reads or writes the wrong buffer (bridged inputs included: `bridged_input_count`).
python tools/conv_fusion_fuzz.py [n_graphs]   (FUZZ_SEED in the environment)"""
import os
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
from conftest import load_backend_module  # noqa: E402

B = load_backend_module()
assert B is not None and hasattr(B, "RocmRuntime"), "plugin build missing"
rocm = B.RocmRuntime(0)
F16 = 10
n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(os.environ.get("FUZZ_SEED", "9"))
bad = 0
TOTAL = [0, 0, 0]


def build(seed, h, feeds):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([2, 4, 8]))
    hw = int(rng.choice([14, 28, 56]))
    cin = int(rng.choice([32, 64, 128]))

    def weight(shape, std):
        t = h.tensor(list(shape), F16)
        t.set_weight()
        feeds.append((t, (rng.standard_normal(shape) * std).astype(np.float16)))
        return t

    def conv_bias(x, ci, co, k, stride, relu):
        w = weight((co, ci, k, k), np.sqrt(2.0 / (ci * k * k)))
        if rng.random() < 0.7:  # the ONNX front-end's form: conv -> reshape(bias, [1, F, 1, 1]) -> add (onnx.py:159-190)
            b = h.reshape(weight((co,), 0.1), None, [1, co, 1, 1])
        else:
            b = weight((1, co, 1, 1), 0.1)
        y = h.add(h.conv(x, w, None, k // 2, k // 2, stride, stride, 1, 1), b, None)
        return h.relu(y, None) if relu else y

    x = h.tensor([n, cin, hw, hw], F16)
    feeds.append((x, rng.uniform(0, 1, (n, cin, hw, hw)).astype(np.float16)))
    y = h.relu(x, None)
    c = cin
    for blk in range(int(rng.integers(1, 4))):
        width = int(rng.choice([32, 64]))
        stride = int(rng.choice([1, 1, 2])) if hw >= 14 else 1
        first = blk == 0 or stride == 2 or c != width * 4
        idt = y
        o = conv_bias(y, c, width, 1, 1, True)
        o = conv_bias(o, width, width, 3, stride, True)
        o = conv_bias(o, width, width * 4, 1, 1, False)
        if first:
            idt = conv_bias(y, c, width * 4, 1, stride, False)
        y = h.relu(h.add(o, idt, None), None)
        c = width * 4
        hw = (hw + 2 - 3) // stride + 1 if stride == 2 else hw
    if rng.random() < 0.5 and hw >= 4:
        y = h.maxPool(y, None, 3, 3, 1, 1, 1, 1, 2, 2, 0)
        hw = (hw + 2 - 3) // 2 + 1
    y = h.avgPool(y, None, hw, hw, 1, 1, 0, 0, 1, 1, 0)
    return [h.flatten(y, None, 1)]


for g in range(n_graphs):
    seed = seed0 * 1000 + g
    got, counts = {}, {}
    try:
        for on in (True, False):
            rocm.set_fusion(on)
            h = B.GraphHandler(rocm)
            feeds = []
            outs = build(seed, h, feeds)
            h.data_malloc()
            for t, a in feeds:
                t.copyin_numpy(np.ascontiguousarray(a))
            c0 = (rocm.fused_launch_count(), rocm.bridged_input_count(), rocm.forwarded_output_count())
            if g % 2:
                h.run_with_hipgraph()
            else:
                h.run()
            counts[on] = (rocm.fused_launch_count() - c0[0], rocm.bridged_input_count() - c0[1], rocm.forwarded_output_count() - c0[2])
            got[on] = [o.copyout_numpy().astype(np.float64) for o in outs]
    finally:
        rocm.set_fusion(True)
    TOTAL[0] += counts[True][0]
    TOTAL[1] += counts[True][1]
    TOTAL[2] += counts[True][2]
    a, b = got[True][0], got[False][0]
    scale = max(1e-6, float(np.abs(b).max()))
    err = float(np.abs(a - b).max()) / scale
    if not (np.isfinite(a).all() and err <= 1.5e-2):
        bad += 1
        print(f"FAIL graph seed {seed}: max diff {err:.3g} of the output scale {scale:.3g}; fused launches {counts[True][0]}, bridged {counts[True][1]}", flush=True)
print(f"{n_graphs - bad}/{n_graphs} graphs agree with fusion on / off ({TOTAL[0]} fused launches, {TOTAL[1]} bridged conv inputs, "
      f"{TOTAL[2]} forwarded outputs)")
sys.exit(1 if bad else 0)
