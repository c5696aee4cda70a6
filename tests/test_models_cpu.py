"""Graph builders of tools/model_bench.py (BASELINE configs 3, 4, 5) checked WITHOUT a GPU: the graphs are built with the
reference's GraphHandler on its native-CPU runtime (shape inference only, nothing runs) — operator counts, output shapes,
and the tensor-parallel sharding of the Llama block (per-rank weight shapes, AllReduce placement)."""
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))


@pytest.fixture(scope="module")
def B(ref_backend):
    return ref_backend


def _ops(h):
    return [str(o).split("(")[0].strip() for o in h.operators()]


def test_resnet50_graph_shape(B):
    from model_bench import Builder, build_resnet50

    bl = Builder(B, B.cpu_runtime(), "f32", seed=0)
    out = build_resnet50(bl, 2, 64)
    assert out.shape() == [2, 1000]
    ops = bl.h.operators()
    # 53 conv + 53 bias Reshapes (onnx.py:159-190) + 53 bias adds + 49 relu + 16 residual adds + 2 pools + flatten + Gemm
    assert len(ops) == 228
    bl2 = Builder(B, B.cpu_runtime(), "f32", seed=0)
    build_resnet50(bl2, 2, 64, frontend=False)
    assert len(bl2.h.operators()) == 175  # the idealised lowering of rounds 1-2 (pre-shaped bias): kept for A/B
    macs = bl.flops / 2 / 2  # per image
    assert abs(macs - 4.09e9 * (64 / 224) ** 2) / (4.09e9 * (64 / 224) ** 2) < 0.08  # ~4.1 GMAC per 224^2 image


def test_bert_graph_shape(B):
    from model_bench import Builder, build_bert

    bl = Builder(B, B.cpu_runtime(), "f32", seed=0)
    out = build_bert(bl, 2, 16, 2, hidden=64, heads=2, ffn=128, vocab=100)
    assert out.shape() == [2, 16, 64]
    # front-end form, per layer: 4 linear (MatMul + Add each) + 2 attention matmuls, 4 reshapes, 4 head transposes + Transpose(K),
    # div, mask add, 2 residual adds, softmax, gelu, 2 layer norms
    assert len(bl.h.operators()) == 3 + 2 * 31
    bl2 = Builder(B, B.cpu_runtime(), "f32", seed=0)
    build_bert(bl2, 2, 16, 2, hidden=64, heads=2, ffn=128, vocab=100, frontend=False)
    assert len(bl2.h.operators()) == 3 + 2 * 24  # idealised: bias inside the MatMul, transB for K^T
    bl3 = Builder(B, B.cpu_runtime(), "f32", seed=0)
    build_bert(bl3, 2, 16, 2, hidden=64, heads=2, ffn=128, vocab=100, decomposed=True)
    # + 8 operators per LayerNorm (5 of them), + 4 per Gelu (2)
    assert len(bl3.h.operators()) == 3 + 2 * 31 + 5 * 8 + 2 * 4


@pytest.mark.parametrize("world", [1, 2, 4])
def test_llama_block_sharding(B, world):
    """Per-rank shapes of the tensor-parallel block (parallel_opt.py rules): column-parallel q/k/v/gate/up keep all rows
    and 1/world of the columns, row-parallel o_proj/down keep 1/world of the rows; exactly two AllReduceSum operators,
    each right after a row-parallel MatMul; the block's output keeps the unsharded shape."""
    from model_bench import Builder, build_llama_block

    heads, D, ffn, S, Bt = 4, 128, 512, 8, 2
    H = heads * D
    # operator-type ids as this build numbers them (OpTypeId does not export the collectives)
    hp = B.GraphHandler(B.cpu_runtime())
    t = hp.tensor([2, 2], 1)
    hp.allReduceSum(hp.matmul(t, t, None, False, False, None, B.ActType.Linear, "default"), None)
    MM, AR = (o.op_type().id() for o in hp.operators())
    for rank in range(world):
        bl = Builder(B, B.cpu_runtime(), "f32", seed=0)
        out = build_llama_block(bl, Bt, S, heads, D, ffn, world, rank)
        assert out.shape() == [Bt, S, H]
        shapes = sorted(tuple(a.shape) for _, a in bl.feeds if a.ndim == 2 and a.dtype == np.float32)
        want = sorted([(H, H // world)] * 3 + [(H // world, H)] + [(H, ffn // world)] * 2 + [(ffn // world, H)])
        assert shapes == want, (world, rank, shapes)
        ids = [o.op_type().id() for o in bl.h.operators()]
        ar = [i for i, t in enumerate(ids) if t == AR]
        assert len(ar) == 2
        for i in ar:
            assert ids[i - 1] == MM
    # different ranks hold different shards of the SAME full weights
    if world > 1:
        b0 = Builder(B, B.cpu_runtime(), "f32", seed=0)
        build_llama_block(b0, Bt, S, heads, D, ffn, world, 0)
        b1 = Builder(B, B.cpu_runtime(), "f32", seed=0)
        build_llama_block(b1, Bt, S, heads, D, ffn, world, 1)
        w0 = [a for _, a in b0.feeds if a.ndim == 2 and a.shape == (H, H // world)]
        w1 = [a for _, a in b1.feeds if a.ndim == 2 and a.shape == (H, H // world)]
        assert not np.array_equal(w0[0], w1[0])
        bf = Builder(B, B.cpu_runtime(), "f32", seed=0)
        build_llama_block(bf, Bt, S, heads, D, ffn, 1, 0)
        wq_full = next(a for _, a in bf.feeds if a.ndim == 2 and a.shape == (H, H))
        assert np.array_equal(np.concatenate([w0[0], w1[0]], 1), wq_full[:, : 2 * H // world])
