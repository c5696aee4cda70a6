"""A graph in which the reference's memory planner (best fit, lowest address: src/core/lazy_allocator.cc:73-125) hands the
Conv operator's OWN output block to another branch's operator that runs between the conv's bias Add and the chain's Relu —
the round-3 advisor finding against buffer forwarding (rocm_fusion.cc: the fused kernel writes the conv's buffer at the
chain's slot, i.e. AFTER that operator wrote its result there). Shared by the CPU dry-run plan test and the GPU numerics test.

Operator list and planned placement (S = bytes of one [n, 32, 8, 8] f16 tensor; x is [n, 64, 8, 8] = 2 S):
  0 Reshape(b)     r   tiny
  1 Sigmoid(q1)    d   [0, S)
  2 Relu(inp)      x   [S, 3S)
  3 Sigmoid(q2)    d2  [3S, 4S)
  4 Sigmoid(q3)    d3  [4S, 5S)
  5 Add(d, d3)     e   [5S, 6S)     d, d3 die
  6 Conv(x, w)     y   [0, S)       (d's block: smallest fit, lowest address); x dies
  7 Add(y, r)      t   [4S, 5S)     (d3's block: exact fit beats x's 2 S block); y dies
  8 Tanh(d2)       v   [0, S)       = y's block; d2 dies
  9 Relu(t)        out [S, 2S)      on x: the chain Conv -> Reshape(bias) -> Add -> Relu cannot write its planned output
 10 Add(out, v)    f1              reads v AFTER the chain's slot
 11 Add(f1, e)     f2              (graph output)
"""
import numpy as np

F16 = 10


def build(B, runtime, n=2, seed=0):
    h = B.GraphHandler(runtime)
    rng = np.random.default_rng(seed)
    t = {}
    for name, shape in (("inp", [n, 64, 8, 8]), ("q1", [n, 32, 8, 8]), ("q2", [n, 32, 8, 8]), ("q3", [n, 32, 8, 8])):
        t[name] = h.tensor(shape, F16)
        t[name].set_input()
    t["w"] = h.tensor([32, 64, 1, 1], F16)
    t["w"].set_weight()
    t["b"] = h.tensor([32], F16)
    t["b"].set_weight()
    t["r"] = h.reshape(t["b"], None, [1, 32, 1, 1])
    t["d"] = h.sigmoid(t["q1"], None)
    t["x"] = h.relu(t["inp"], None)
    t["d2"] = h.sigmoid(t["q2"], None)
    t["d3"] = h.sigmoid(t["q3"], None)
    t["e"] = h.add(t["d"], t["d3"], None)
    t["y"] = h.conv(t["x"], t["w"], None, 0, 0, 1, 1, 1, 1)
    t["t"] = h.add(t["y"], t["r"], None)
    t["v"] = h.tanh(t["d2"], None)
    t["out"] = h.relu(t["t"], None)
    t["f1"] = h.add(t["out"], t["v"], None)
    t["f2"] = h.add(t["f1"], t["e"], None)
    h.data_malloc()
    feeds = {
        "inp": rng.standard_normal((n, 64, 8, 8)).astype(np.float16),
        "q1": rng.standard_normal((n, 32, 8, 8)).astype(np.float16),
        "q2": rng.standard_normal((n, 32, 8, 8)).astype(np.float16),
        "q3": rng.standard_normal((n, 32, 8, 8)).astype(np.float16),
        "w": (rng.standard_normal((32, 64, 1, 1)) / 8).astype(np.float16),
        "b": rng.standard_normal((32,)).astype(np.float16),
    }
    return h, t, feeds


def oracle(feeds):
    f = {k: v.astype(np.float64) for k, v in feeds.items()}
    r16 = lambda a: a.astype(np.float16).astype(np.float64)
    sig = lambda a: r16(1 / (1 + np.exp(-a)))
    d, d2, d3 = sig(f["q1"]), sig(f["q2"]), sig(f["q3"])
    x = np.maximum(f["inp"], 0)
    e = r16(d + d3)
    y = r16(np.einsum("nchw,fc->nfhw", x, f["w"][:, :, 0, 0]))
    t = r16(y + f["b"].reshape(1, -1, 1, 1))
    v = r16(np.tanh(d2))
    out = np.maximum(t, 0)
    return r16(r16(out + v) + e)
