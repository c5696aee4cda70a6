#!/usr/bin/env python3
"""Tensor-parallel launcher for Device::ROCM — the MI355X peer of the reference's
examples/distributed/cuda/cuda_launch.py (one process per GPU, `runtime.init_comm(name, world, rank)` with the file
rendezvous, the graph run through the reference executor, the single-GPU result as the standard, "max abs diff"
printed per rank, cuda_launch.py:64-108).

The reference launcher shards an ONNX file with parallel_opt.parallel_model; onnx / onnxsim are not installed on this
image, so the model is the Llama-7B-style decoder block of BASELINE config 5 built op by op with backend.GraphHandler
(tools/model_bench.py::build_llama_block) and sharded by the same rules (infinitensor_amd/tp.py): q/k/v/gate/up
column-parallel, o_proj/down row-parallel, one AllReduceSum (RCCL over xGMI, on the runtime stream, captured in the
hipGraph) after each row-parallel MatMul.

  python tools/rocm_launch.py --nproc_per_node 8 [--batch_size 4] [--length 512] [--type fp16] [--gen_std]

Step 1 (`--gen_std`, or automatically when the standard is missing): the unsharded block on GPU 0 -> <name>_results.npy.
Step 2: `nproc_per_node` workers, each prints its max abs diff against the standard and the block latency; rank 0 prints
one JSON line with the max over ranks.
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tools"))
sys.path.insert(0, str(REPO / "tests"))


def parse_args():
    ap = argparse.ArgumentParser(description="launch a tensor-parallel InfiniTensor graph on MI355X GPUs")
    ap.add_argument("--nproc_per_node", type=int, default=1)
    ap.add_argument("--name", type=str, default="llama_block")
    ap.add_argument("--batch_size", type=int, default=4)
    ap.add_argument("--length", type=int, default=512)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--ffn", type=int, default=11008)
    ap.add_argument("--type", choices=["fp32", "fp16"], default="fp16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--gen_std", action="store_true", help="only generate the single-GPU standard result")
    return ap.parse_args()


def run_block(args, world: int, rank: int, local_rank: int, comm_name: str):
    """Build this rank's shard of the block, run it eagerly once (the result), then time hipGraph replays."""
    from model_bench import Builder, build_llama_block, load_backend, timed

    B = load_backend()
    rt = B.RocmRuntime(local_rank)
    rt.init_comm(comm_name, world, rank)
    bl = Builder(B, rt, {"fp32": "f32", "fp16": "f16"}[args.type], seed=0)
    out = build_llama_block(bl, args.batch_size, args.length, args.heads, 128, args.ffn, world, rank)
    bl.finish()
    bl.h.run()
    y = out.copyout_numpy().astype(np.float32)
    ms = timed(bl.h.run_with_hipgraph, args.iters)
    return y, ms, bl.flops


def gen_standard(args, q=None):
    y, ms, _ = run_block(args, 1, 0, 0, args.name + "_std")
    np.save(f"{args.name}_results.npy", y)
    print(f"standard: outputs abs mean {np.abs(y).mean():.6f}, {ms:.3f} ms per block on one GPU", flush=True)
    if q is not None:
        q.put(ms)


def start_worker(args, world: int, rank: int, q):
    y, ms, flops = run_block(args, world, rank, rank, args.name + "_dist")
    std = np.load(f"{args.name}_results.npy")
    diff = float(np.abs(y - std).max())
    print(f"rank {rank}: outputs abs mean {np.abs(y).mean():.6f}  max abs diff: {diff:.3e}  {ms:.3f} ms per block", flush=True)
    q.put((rank, diff, ms, flops, bool(np.isfinite(y).all())))


def main():
    args = parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ctx = mp.get_context("spawn")  # a fresh HIP context per process (the reference isolates CUDA the same way)
    world = args.nproc_per_node
    single_ms = None
    if args.gen_std or not Path(f"{args.name}_results.npy").exists():
        q = ctx.Queue()
        p = ctx.Process(target=gen_standard, args=(args, q))
        p.start()
        p.join()
        if p.exitcode != 0:
            raise SystemExit(f"generating the standard failed (exit {p.exitcode})")
        single_ms = q.get()
        if args.gen_std:
            return
    q = ctx.Queue()
    procs = [ctx.Process(target=start_worker, args=(args, world, r, q)) for r in range(world)]
    t0 = time.time()
    for p in procs:
        p.start()
    for p in procs:
        p.join()
    bad = [p.exitcode for p in procs if p.exitcode != 0]
    if bad:
        raise SystemExit(f"workers failed: exit codes {bad}")
    res = sorted(q.get() for _ in range(world))
    ms = max(r[2] for r in res)
    tokens = args.batch_size * args.length
    print(json.dumps({"workload": f"Llama-7B-style block, {tokens} tokens, {args.type}, TP={world}, reference executor + ROCM plugin, hipGraph",
                      "ms_per_block": round(ms, 4), "single_gpu_ms": None if single_ms is None else round(single_ms, 4),
                      "gemm_TFLOPs_aggregate": round(world * res[0][3] / ms / 1e9, 1),
                      "max_abs_diff_vs_single_gpu": max(r[1] for r in res), "finite": all(r[4] for r in res),
                      "wall_s": round(time.time() - t0, 1)}))


if __name__ == "__main__":
    main()
