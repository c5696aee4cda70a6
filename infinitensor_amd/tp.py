"""Megatron-style tensor-parallel sharding helpers for the transformer-block path (host logic only).

Mirrors what the reference does as an ONNX graph rewrite in examples/distributed/parallel_opt.py:
  * column-parallel first linear: weight [in, out] sharded on the LAST dim, bias sharded with it,
    output sharded, no communication (parallel_opt.py:46-59 `shard_gemm`, :71-79);
  * row-parallel second linear: weight sharded on dim 0, output is a partial sum -> ONE AllReduceSum,
    bias added once after the reduce (parallel_opt.py:195-224);
  * attention heads split by the Reshape rewrite (parallel_opt.py:81-119): here `shard_heads`.
Array-library agnostic (numpy arrays or torch tensors: only slicing is used).
"""
from __future__ import annotations


def shard_range(n: int, world: int, rank: int) -> tuple[int, int]:
    if n % world != 0:
        raise ValueError(f"dimension {n} is not divisible by world size {world}")
    step = n // world
    return rank * step, (rank + 1) * step


def shard_column(w, world: int, rank: int, bias=None):
    """weight [in, out] -> [in, out/world]; bias [out] -> [out/world]."""
    lo, hi = shard_range(w.shape[-1], world, rank)
    return (w[..., lo:hi], None if bias is None else bias[lo:hi])


def shard_row(w, world: int, rank: int):
    """weight [in, out] -> [in/world, out]; the matching activation is the column-parallel output."""
    lo, hi = shard_range(w.shape[0], world, rank)
    return w[lo:hi]


def shard_heads(wqkv, n_heads: int, head_dim: int, world: int, rank: int):
    """fused qkv weight [in, 3*n_heads*head_dim] (q | k | v blocks) -> this rank's heads of each block."""
    hidden = n_heads * head_dim
    lo, hi = shard_range(n_heads, world, rank)
    parts = [wqkv[..., i * hidden + lo * head_dim: i * hidden + hi * head_dim] for i in range(3)]
    return parts


def llama_block_flops(tokens: int, hidden: int, ffn: int, world: int) -> float:
    """per-rank GEMM FLOP of one decoder block: 2*T*(4*H^2 + 3*H*F)/tp."""
    return 2.0 * tokens * (4.0 * hidden * hidden + 3.0 * hidden * ffn) / world
