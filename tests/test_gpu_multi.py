"""N > 1 on hardware: the RCCL path of SURVEY 8a row a11 / 8f-2 with one process per GPU.

Collected everywhere, skipped on boxes with fewer GPUs than the case needs (the builder's `gpurun` boxes have one;
the world-1 case runs there and keeps the worker script itself honest).
Each case starts `world` workers (tests/_rccl_worker.py; RANK / WORLD_SIZE in the environment, file rendezvous in a
temporary cwd like the reference's NcclCommunicatorObj) which run the reference's collective tests on Device::ROCM
through both the C ABI and the reference executor + plugin:
test/kernels/cuda/test_cuda_all_reduce.cc:38-106, test_cuda_all_gather.cc:38-50, test_cuda_broadcast.cc:41-55,
test_cuda_sendrecv.cc:50-87 (worlds 3 and 4 there), test/cuda/test_nccl_comm.cc:37-52.
Plus the 2-rank launcher check of examples/distributed/cuda/cuda_launch.py:70-76 (max-abs-diff of the tensor-parallel
Llama block against the single-GPU result) and `bench.py --gpus 2` spawning its own ranks."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent
NGPU = torch.cuda.device_count() if torch.cuda.is_available() else 0


def need(n):
    return pytest.mark.skipif(NGPU < n, reason=f"needs {n} GPUs on one node (have {NGPU})")


def launch(world: int, script: Path, cwd: Path, extra_args=(), timeout=600, extra_env=None):
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, str(script), *extra_args], env=env, cwd=cwd, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed (rc {p.returncode}):\n{se[-3000:]}"
    return [so for so, _ in outs]


@pytest.mark.parametrize("world", [1, pytest.param(2, marks=need(2)), pytest.param(3, marks=need(3)), pytest.param(4, marks=need(4)),
                                   pytest.param(8, marks=need(8))])
def test_rccl_collectives_match_the_reference_tests(world, tmp_path):
    outs = launch(world, REPO / "tests" / "_rccl_worker.py", tmp_path)
    for r, so in enumerate(outs):
        res = json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT ")][-1][7:])
        assert res["rank"] == r and res["world"] == world
        assert set(res["done"]) >= {"abi_all_reduce", "abi_all_gather", "abi_broadcast", "abi_send_recv",
                                    "abi_all_reduce_hipgraph", "plugin_collectives", "plugin_all_reduce_hipgraph",
                                    "abi_all_reduce_overlapped", "abi_reduce_scatter", "plugin_row_parallel_overlap"}


@pytest.mark.parametrize("world", [2, 4, 8])
def test_direct_transport_collectives_on_one_device(world, tmp_path):
    """The hand-written IPC / xGMI transport (csrc/comm_direct.hip; INFINI_ROCM_COMM=direct) with `world` ranks that all open
    device 0 — RCCL refuses two ranks per device, push kernels over IPC-mapped buffers do not — running the reference's
    collective tests (test_cuda_all_reduce.cc:38-106, test_cuda_all_gather.cc:38-50, test_cuda_broadcast.cc:41-55,
    test_cuda_sendrecv.cc:50-87, test_nccl_comm.cc:37-52) through the C ABI and through the reference executor + plugin, plus
    the transport's own stress cases (bit-exact integer sums vs the oracle, multi-piece messages, odd counts, back-to-back
    calls, broadcast roots in a row, a send / recv ring): the first non-identity reduction evidence on a one-GPU box."""
    # world 2 also runs the chunked, overlapped MatMul -> AllReduceSum plan (INFINI_ROCM_TP_OVERLAP=1: opt-in since round 6); worlds 4
    # and 8 run the default plan, whose assertion in the worker is ONE all-reduce per row-parallel GEMM at this size
    extra = {"INFINI_ROCM_COMM": "direct", "IROCM_WORKER_SHARED_DEVICE": "1", "INFINI_ROCM_DIRECT_TIMEOUT_S": "60"}
    if world == 2:
        extra["INFINI_ROCM_TP_OVERLAP"] = "1"
    outs = launch(world, REPO / "tests" / "_rccl_worker.py", tmp_path, timeout=900, extra_env=extra)
    for r, so in enumerate(outs):
        res = json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT ")][-1][7:])
        assert res["rank"] == r and res["world"] == world
        assert set(res["done"]) >= {"abi_all_reduce", "abi_all_gather", "abi_broadcast", "abi_send_recv", "abi_all_reduce_hipgraph",
                                    "plugin_collectives", "plugin_all_reduce_hipgraph", "abi_all_reduce_overlapped", "abi_reduce_scatter",
                                    "plugin_row_parallel_overlap", "direct_stress"}


def test_row_parallel_overlap_forced_on_one_rank(tmp_path):
    """The chunked, overlapped MatMul -> AllReduceSum launch only plans itself with more than one rank; forced here at
    world 1 so that the code path (comm stream, fork / join events, capture) runs on the builder's one-GPU boxes too."""
    outs = launch(1, REPO / "tests" / "_rccl_worker.py", tmp_path, extra_env={"INFINI_ROCM_TP_OVERLAP": "force"})
    res = json.loads([ln for ln in outs[0].splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert "plugin_row_parallel_overlap" in res["done"]


@need(2)
def test_rocm_launch_two_ranks_match_single_gpu(tmp_path):
    """cuda_launch.py:70-76: the sharded graph's output vs the single-GPU standard; fp16 tolerance."""
    r = subprocess.run([sys.executable, str(REPO / "tools" / "rocm_launch.py"), "--nproc_per_node", "2", "--iters", "5"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "TP=2" in line["workload"] and line["max_abs_diff_vs_single_gpu"] < 5e-2 and line["finite"]


@need(2)
def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` (no launcher): n_gpus 2, a real all-reduce bus bandwidth, TP parity."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--no-graph", "--print-detail"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0
    tp = line["tp_block"]
    # the flat scalars the driver's record keeps (config.*) carry the same figures as the nested detail
    assert line["config"]["allreduce_16MiB_rccl_busbw_GBs"] == tp["allreduce_busbw_GBs"] and line["config"]["llama7b_block_tp2_ms"] == tp["ms_per_block"]
    assert tp["allreduce_busbw_GBs"] and tp["allreduce_busbw_GBs"] > 1.0
    assert tp["max_abs_diff_vs_unsharded"] < 5e-2 and tp["finite"]


@pytest.mark.parametrize("world", [2, 4])
def test_bench_tp_block_on_one_device_over_the_direct_transport(world):
    """`bench.py --gpus N` with its N ranks on ONE device (IROCM_BENCH_ONE_DEVICE=1, INFINI_ROCM_COMM=direct, gloo as
    torch.distributed's backend): the Llama-7B tensor-parallel block of BASELINE config 5 — column / row sharded GEMMs, two
    16 MiB all-reduces on the hand-written transport, the chunked overlap form, reduce-scatter + all-gather — must reproduce
    the UNSHARDED block (the reference's own check, cuda_launch.py:70-76). The timings of such a run mean nothing (the ranks
    share a GPU); the parity does, and it is the first one measured at world > 1."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(IROCM_BENCH_ONE_DEVICE="1", INFINI_ROCM_COMM="direct", HSA_ENABLE_IPC_MODE_LEGACY="0", INFINI_ROCM_DIRECT_TIMEOUT_S="60",
               IROCM_BENCH_TP_DEBUG="1")
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", str(world), "--steps", "5", "--warmup", "2", "--no-graph",
                        "--no-cpu-baseline", "--no-extras", "--print-detail"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == world and line["value"] > 0
    tp = line["tp_block"]
    assert line["config"][f"llama7b_block_tp{world}_ms"] == tp["ms_per_block"]
    assert f"TP={world}" in tp["workload"]
    dbg = [ln for ln in r.stderr.splitlines() if "[tp debug]" in ln]
    assert tp["finite"] and tp["max_abs_diff_vs_unsharded"] < 5e-2, (tp, dbg)
    assert tp["allreduce_16MiB_ms"] and tp["allreduce_16MiB_ms"] > 0
    assert tp["overlap"]["max_abs_diff_on_vs_off"] < 5e-2
    assert "error" not in (tp["reduce_scatter_all_gather"] or {}), tp["reduce_scatter_all_gather"]
