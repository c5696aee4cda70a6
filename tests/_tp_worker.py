"""Worker of tests/test_distributed_cpu.py: one rank of a world_size-2 gloo job (CPU)."""
import json
import os
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist

    from infinitensor_amd import tp
    from oracle import ref_ops as R

    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(0)  # every rank builds the same full weights, then slices its own shard
    T, H, F, NH = 16, 32, 64, 4
    x = rng.standard_normal((T, H))
    w1, b1 = rng.standard_normal((H, F)), rng.standard_normal(F)
    w2, b2 = rng.standard_normal((F, H)), rng.standard_normal(H)
    wqkv = rng.standard_normal((H, 3 * H))
    wo = rng.standard_normal((H, H))
    # MLP: column-parallel -> gelu -> row-parallel -> all-reduce -> bias once
    w1s, b1s = tp.shard_column(w1, world, rank, b1)
    h = R.unary("gelu", R.matmul(x, w1s, b1s))
    part = torch.from_numpy(np.ascontiguousarray(R.matmul(h, tp.shard_row(w2, world, rank))))
    dist.all_reduce(part, op=dist.ReduceOp.SUM)
    mlp = part.numpy() + b2
    # attention: heads sharded (q, k, v column-parallel), output projection row-parallel
    D = H // NH
    q, k, v = (R.matmul(x, w) for w in tp.shard_heads(wqkv, NH, D, world, rank))
    nh = NH // world
    heads = lambda y: y.reshape(T, nh, D).transpose(1, 0, 2)
    ctx = R.attention(heads(q), heads(k), heads(v), 1.0 / np.sqrt(D)).transpose(1, 0, 2).reshape(T, nh * D)
    part = torch.from_numpy(np.ascontiguousarray(R.matmul(ctx, tp.shard_row(wo, world, rank))))
    dist.all_reduce(part, op=dist.ReduceOp.SUM)
    attn = part.numpy()
    if rank == 0:
        full_mlp = R.matmul(R.unary("gelu", R.matmul(x, w1, b1)), w2, b2)
        qf, kf, vf = (R.matmul(x, wqkv[:, i * H:(i + 1) * H]) for i in range(3))
        hf = lambda y: y.reshape(T, NH, D).transpose(1, 0, 2)
        full_attn = R.matmul(R.attention(hf(qf), hf(kf), hf(vf), 1.0 / np.sqrt(D)).transpose(1, 0, 2).reshape(T, H), wo)
        print("RESULT " + json.dumps({"mlp": float(np.abs(mlp - full_mlp).max()), "attn": float(np.abs(attn - full_attn).max())}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
