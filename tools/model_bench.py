"""Graph-level benchmarks through the REFERENCE graph executor + the Device::ROCM plugin
(BASELINE configs 3 and 4): the graphs are built op by op with backend.GraphHandler exactly as
OnnxStub would emit them (onnx/onnxsim are not installed here): BN folded -> Conv + Add(bias) + Relu,
MatMul + Add(bias), decomposed attention (MatMul, Div, Add(mask), Softmax, MatMul).

  python tools/model_bench.py resnet50 [--batch 128] [--dtype f16]
  python tools/model_bench.py bert     [--batch 32] [--seq 512] [--layers 12]
  python tools/model_bench.py llama    [--batch 4] [--seq 512]        (config 5 at TP = 1; TP > 1: tools/rocm_launch.py)

Prints one JSON line per model: eager ms/run (host loop + launches + one sync) and hipGraph replay ms/run.
Weights: N(0, sqrt(2/fan_in)) / N(0, 0.02), seed 0; inputs seeded. The parity of these graphs is
covered by tests (tests/test_gpu_plugin.py, tests/test_gpu_models.py) on small slices.
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))

DT = {"f32": (1, np.float32), "f16": (10, np.float16)}


def load_backend():
    from conftest import load_backend_module

    b = load_backend_module()
    if b is None or not hasattr(b, "RocmRuntime"):
        raise SystemExit("plugin build missing: run __graft_entry__.build() where /root/reference exists")
    return b


class Builder:
    """Tiny helper that creates weight tensors and remembers what to copy in after data_malloc."""

    def __init__(self, B, rt, dtype: str, seed: int = 0):
        self.B, self.h = B, B.GraphHandler(rt)
        self.code, self.np = DT[dtype]
        self.rng = np.random.default_rng(seed)
        self.feeds = []
        self.flops = 0.0

    def weight(self, shape, std):
        t = self.h.tensor(list(shape), self.code)
        t.set_weight()
        self.feeds.append((t, (self.rng.standard_normal(shape) * std).astype(self.np)))
        return t

    def const(self, arr, code=None):
        t = self.h.tensor(list(arr.shape), code or self.code)
        t.set_weight()
        self.feeds.append((t, arr))
        return t

    def input(self, arr, code=None):
        t = self.h.tensor(list(arr.shape), code or self.code)
        t.set_input()
        self.feeds.append((t, arr))
        return t

    def finish(self):
        self.h.data_malloc()
        for t, a in self.feeds:
            t.copyin_numpy(np.ascontiguousarray(a))


def build_resnet50(bl: Builder, batch: int, image: int = 224, fc_bias_as_add: bool = False):
    """fc_bias_as_add: emit the classifier as MatMul + Add (the reference's native-CPU MatMul has no bias input), so the
    same graph also runs on `backend.cpu_runtime()` for end-to-end parity (tests/test_gpu_models.py)."""
    h = bl.h

    def conv_bn_act(x, cin, cout, k, stride, pad, relu=True, hw=None):
        w = bl.weight((cout, cin, k, k), np.sqrt(2.0 / (cin * k * k)))
        b = bl.weight((1, cout, 1, 1), 0.01)
        y = h.conv(x, w, None, pad, pad, stride, stride, 1, 1)
        oh = (hw + 2 * pad - k) // stride + 1
        bl.flops += 2.0 * batch * cout * oh * oh * cin * k * k
        y = h.add(y, b, None)
        return (h.relu(y, None) if relu else y), oh

    x = bl.input(bl.rng.uniform(0, 1, (batch, 3, image, image)).astype(bl.np))
    y, hw = conv_bn_act(x, 3, 64, 7, 2, 3, hw=image)
    y = h.maxPool(y, None, 3, 3, 1, 1, 1, 1, 2, 2, 0)
    hw = (hw + 2 - 3) // 2 + 1
    cin = 64
    for width, blocks, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)):
        for bi in range(blocks):
            s = stride if bi == 0 else 1
            idt = y
            o, hw1 = conv_bn_act(y, cin, width, 1, 1, 0, hw=hw)
            o, hw2 = conv_bn_act(o, width, width, 3, s, 1, hw=hw1)
            o, hw3 = conv_bn_act(o, width, width * 4, 1, 1, 0, relu=False, hw=hw2)
            if bi == 0:
                idt, _ = conv_bn_act(y, cin, width * 4, 1, s, 0, relu=False, hw=hw)
            y = h.relu(h.add(o, idt, None), None)
            cin, hw = width * 4, hw3
    y = h.avgPool(y, None, hw, hw, 1, 1, 0, 0, 1, 1, 0)
    y = h.flatten(y, None, 1)
    wfc = bl.weight((2048, 1000), np.sqrt(1.0 / 2048))
    bfc = bl.weight((1000,), 0.01)
    if fc_bias_as_add:
        y = h.add(h.matmul(y, wfc, None, False, False, None, bl.B.ActType.Linear, "default"), bfc, None)
    else:
        y = h.matmul(y, wfc, None, False, False, bfc, bl.B.ActType.Linear, "default")
    bl.flops += 2.0 * batch * 2048 * 1000
    return y


def build_bert(bl: Builder, batch: int, seq: int, layers: int, hidden: int = 768, heads: int = 12, ffn: int = 3072,
               vocab: int = 30522):
    h, B = bl.h, bl.B
    lin = B.ActType.Linear
    D = hidden // heads
    ids = bl.input(bl.rng.integers(0, vocab, (batch, seq)).astype(np.int64), 7)
    emb = bl.weight((vocab, hidden), 0.02)
    pos = bl.weight((1, seq, hidden), 0.02)
    mask = bl.const(np.zeros((batch, 1, 1, seq), bl.np))
    scale = bl.const(np.array([np.sqrt(D)], bl.np))
    x = h.add(h.gather(emb, ids, None, 0), pos, None)

    def ln(t):
        return h.layerNormalization(t, bl.const(np.ones(hidden, bl.np)), None, bl.const(np.zeros(hidden, bl.np)), 1e-12, 2, 1)

    def linear(t, cin, cout):
        w = bl.weight((cin, cout), 0.02)
        b = bl.weight((cout,), 0.02)
        bl.flops += 2.0 * batch * seq * cin * cout
        return h.matmul(t, w, None, False, False, b, lin, "default")

    x = ln(x)
    for _ in range(layers):
        def heads_of(t):
            return h.transpose(h.reshape(t, None, [batch, seq, heads, D]), None, [0, 2, 1, 3])
        q, k, v = heads_of(linear(x, hidden, hidden)), heads_of(linear(x, hidden, hidden)), heads_of(linear(x, hidden, hidden))
        s = h.matmul(q, k, None, False, True, None, lin, "default")
        bl.flops += 2.0 * batch * heads * seq * seq * D * 2
        s = h.add(h.div(s, scale, None), mask, None)
        p = h.softmax(s, None, 3)
        ctx = h.matmul(p, v, None, False, False, None, lin, "default")
        ctx = h.reshape(h.transpose(ctx, None, [0, 2, 1, 3]), None, [batch, seq, hidden])
        x = ln(h.add(x, linear(ctx, hidden, hidden), None))
        f = linear(h.gelu(linear(x, hidden, ffn), None), ffn, hidden)
        x = ln(h.add(x, f, None))
    return x


def build_llama_block(bl: Builder, batch: int, seq: int, heads: int = 32, head_dim: int = 128, ffn: int = 11008,
                      world: int = 1, rank: int = 0, all_reduce: bool = True):
    """BASELINE config 5: one Llama-7B-style decoder block, tensor-parallel over `world` ranks the way
    examples/distributed/parallel_opt.py rewrites it — q/k/v/gate/up column-parallel (weight sharded on the last dim,
    heads split), o_proj/down row-parallel (weight sharded on dim 0) followed by ONE AllReduceSum each
    (parallel_opt.py:46-59,81-119,195-210). Every rank draws the same full weights (seeded) and keeps its shard
    (infinitensor_amd/tp.py). Attention is the decomposed chain with an additive causal mask [1, 1, S, S].
    all_reduce=False leaves the two partial sums un-reduced (their sum over ranks must equal the unsharded block:
    tests/test_gpu_models.py)."""
    from infinitensor_amd import tp

    h, B = bl.h, bl.B
    lin = B.ActType.Linear
    H = heads * head_dim
    nh = heads // world
    T = batch * seq
    x = bl.input((bl.rng.standard_normal((batch, seq, H))).astype(bl.np))
    full = {n: (bl.rng.standard_normal(shape) * 0.02).astype(bl.np)
            for n, shape in (("q", (H, H)), ("k", (H, H)), ("v", (H, H)), ("o", (H, H)), ("g", (H, ffn)), ("u", (H, ffn)), ("d", (ffn, H)))}
    n1, n2 = (1 + 0.1 * bl.rng.standard_normal(H)).astype(bl.np), (1 + 0.1 * bl.rng.standard_normal(H)).astype(bl.np)
    col = lambda n: bl.const(np.ascontiguousarray(tp.shard_column(full[n], world, rank)[0]))
    row = lambda n: bl.const(np.ascontiguousarray(tp.shard_row(full[n], world, rank)))
    pos = bl.const(np.tile(np.arange(seq, dtype=np.uint32), (batch, 1)), 12)
    mask = bl.const(np.triu(np.full((seq, seq), -1e4, bl.np), 1).reshape(1, 1, seq, seq))
    scale = bl.const(np.array([np.sqrt(head_dim)], bl.np))
    mm = lambda a, w: h.matmul(a, w, None, False, False, None, lin, "default")
    hd = lambda t: h.transpose(h.reshape(t, None, [batch, seq, nh, head_dim]), None, [0, 2, 1, 3])
    hn = h.RMSNorm(x, bl.const(n1), None)
    q, k = hd(h.RoPE(pos, mm(hn, col("q")), None)), hd(h.RoPE(pos, mm(hn, col("k")), None))
    v = hd(mm(hn, col("v")))
    s = h.add(h.div(h.matmul(q, k, None, False, True, None, lin, "default"), scale, None), mask, None)
    ctx = h.matmul(h.softmax(s, None, 3), v, None, False, False, None, lin, "default")
    ctx = h.reshape(h.transpose(ctx, None, [0, 2, 1, 3]), None, [batch, seq, nh * head_dim])
    o = mm(ctx, row("o"))
    if all_reduce:
        o = h.allReduceSum(o, None)
        x1 = h.add(x, o, None)
    else:
        x1 = x  # partial sums stay separate: the caller adds them up over the ranks
    h2 = h.RMSNorm(x1, bl.const(n2), None)
    d = mm(h.mul(h.silu(mm(h2, col("g")), None), mm(h2, col("u")), None), row("d"))
    bl.flops += tp.llama_block_flops(T, H, ffn, world) + 4.0 * batch * nh * seq * seq * head_dim
    if all_reduce:
        return h.add(x1, h.allReduceSum(d, None), None)
    return o, d


def build_matmul(bl: Builder, n: int = 4096, trans_b: bool = False):
    """One n^3 MatMul as a one-operator graph: what a single launch costs through the reference executor."""
    a = bl.input((bl.rng.standard_normal((n, n))).astype(bl.np))
    w = bl.weight((n, n), 1.0)
    bl.flops += 2.0 * n ** 3
    return bl.h.matmul(a, w, None, False, trans_b, None, bl.B.ActType.Linear, "default")


def timed(fn, iters, warm_ms=40.0):
    # warm by time, not by count: the chip clocks up for ~15 ms from idle (profiles/r02_launch_series.txt)
    t0, n = time.perf_counter(), 0
    while n < 2 or (time.perf_counter() - t0) * 1e3 < warm_ms:
        fn()
        n += 1
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    return (time.perf_counter() - t0) / iters * 1e3


def run_model(model: str, device: int = 0, batch: int | None = None, seq: int = 512, layers: int = 12,
              dtype: str = "f16", iters: int = 10, tune: bool = False) -> dict:
    """tune=True additionally runs the reference's h.tune() (MatMul / Conv pick their kernel variant by measurement,
    plugin/src/rocm_kernels.cc RocmTunableKernel) and times the graph again with the records in the PerfEngine."""
    B = load_backend()
    rt = B.RocmRuntime(device)
    bl = Builder(B, rt, dtype)
    if model == "resnet50":
        batch = batch or 128
        out = build_resnet50(bl, batch)
        name = f"ResNet-50 bs{batch} {dtype}"
    elif model == "llama":
        batch = batch or 4
        rt.init_comm("model_bench_llama", 1, 0)
        out = build_llama_block(bl, batch, seq)
        name = f"Llama-7B block bs{batch} seq{seq} {dtype} TP=1"
    elif model in ("matmul", "matmul_nt"):
        batch = 1
        out = build_matmul(bl, 4096, model == "matmul_nt")
        name = f"MatMul 4096^3 {dtype} " + ("NT" if model == "matmul_nt" else "NN")
    else:
        batch = batch or 32
        out = build_bert(bl, batch, seq, layers)
        name = f"BERT-base L{layers} bs{batch} seq{seq} {dtype}"
    nops = len(bl.h.operators())
    bl.finish()
    f0 = rt.fused_launch_count()
    bl.h.run()
    fused = rt.fused_launch_count() - f0
    eager = timed(bl.h.run, iters)
    graph = timed(bl.h.run_with_hipgraph, iters)
    y = out.copyout_numpy()
    tuned = {}
    if tune:
        import tempfile

        t0 = time.perf_counter()
        bl.h.tune()
        tune_s = time.perf_counter() - t0
        rt.clear_hip_graph_cache()  # the captured launches still carry the heuristic choices
        te = timed(bl.h.run, iters)
        tg = timed(bl.h.run_with_hipgraph, iters)
        with tempfile.TemporaryDirectory() as d:
            B.RocmRuntime.save_perf(d + "/perf.json")
            recs = json.loads(open(d + "/perf.json").read())["data"]
        picks = {}
        for _, r in recs:
            if r["type"] in (3, 4):
                k = ("matmul" if r["type"] == 3 else "conv") + ":" + ("heuristic" if r["data"][0] < 0 else f"variant{r['data'][0]}")
                picks[k] = picks.get(k, 0) + 1
        y = out.copyout_numpy()
        tuned = {"tuned_eager_ms": round(te, 3), "tuned_hipgraph_ms": round(tg, 3), "tune_seconds": round(tune_s, 2),
                 "tuned_picks": picks}
    return {**tuned, "model": name, "ops": nops, "fusion": bool(rt.get_fusion()), "fused_launches_per_run": int(fused), "gemm_conv_TFLOP": round(bl.flops / 1e12, 3),
            "eager_ms": round(eager, 3), "hipgraph_ms": round(graph, 3),
            "hipgraph_TFLOPs": round(bl.flops / graph / 1e9, 1), "batch": batch,
            "per_unit": f"{batch / graph * 1e3:.0f} samples/s", "finite": bool(np.isfinite(y.astype(np.float32)).all())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model", choices=["resnet50", "bert", "llama", "matmul", "matmul_nt"])
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--tune", action="store_true", help="also time the graph after h.tune() (autotuned MatMul / Conv variants)")
    args = ap.parse_args()
    print(json.dumps(run_model(args.model, 0, args.batch, args.seq, args.layers, args.dtype, args.iters, args.tune)))


if __name__ == "__main__":
    main()
