"""Worker of tests/test_gpu_multi.py: one rank (= one process = one GPU) of an RCCL job on one node.

Runs the reference's collective tests on Device::ROCM, twice: through the C ABI (ctypes) and through the reference's
graph executor + plugin (`backend.RocmRuntime(rank).init_comm(name, world, rank)`, file rendezvous in the cwd exactly
like NcclCommunicatorObj): test/kernels/cuda/test_cuda_all_reduce.cc:38-106, test_cuda_all_gather.cc:38-50,
test_cuda_broadcast.cc:41-55, test_cuda_sendrecv.cc:50-87, test/cuda/test_nccl_comm.cc:37-52.
Prints "RESULT {json}" on success; any mismatch raises (non-zero exit)."""
import json
import os
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch

    from infinitensor_amd import RocmRuntime, ops

    torch.cuda.set_device(rank)
    done = []

    # ---- C ABI -------------------------------------------------------------------------------------
    rt = RocmRuntime(rank)
    rt.use_torch_stream()  # order with torch's tensor initialisation
    rt.init_comm("abi_comm", world, rank)  # file rendezvous ./abi_comm_nccl_id.bin (nccl_communicator.h:27-51)
    assert rt.comm_info() == (world, rank)
    dev = lambda a, dt=torch.float32: torch.tensor(a, dtype=dt, device=f"cuda:{rank}")
    # test_nccl_comm.cc:37-52: {1, 4} -> {5, 5}
    y = ops.all_reduce(rt, "sum", dev([1.0 if rank == 0 else 4.0] if world == 2 else [float(rank + 1)]))
    rt.sync()
    assert y.item() == (5.0 if world == 2 else world * (world + 1) / 2)
    # test_cuda_all_reduce.cc:38-106 (ranks >= 2 of a bigger world contribute rank-dependent rows)
    rows = [[2.0, 3.0], [5.0, 6.0]] + [[float(r + 7), float(r + 1)] for r in range(2, world)]
    R = np.array(rows[:world])
    want = {"sum": R.sum(0), "prod": R.prod(0), "min": R.min(0), "max": R.max(0), "avg": R.mean(0)}
    for kind, w in want.items():
        for dt, tol in ((torch.float32, 1e-6), (torch.float16, 2e-3)):
            y = ops.all_reduce(rt, kind, dev(rows[rank], dt))
            rt.sync()
            assert np.allclose(y.float().cpu().numpy(), w, rtol=tol), (kind, dt, y, w)
    done.append("abi_all_reduce")
    # a payload at the size of the TP block's messages: 16 MiB fp16, sum of rank-coloured data
    n = 8 * 1024 * 1024
    big = torch.full((n,), float(rank + 1), dtype=torch.float16, device=f"cuda:{rank}")
    ops.all_reduce(rt, "sum", big, out=big)
    rt.sync()
    assert torch.all(big == world * (world + 1) / 2).item()
    # test_cuda_all_gather.cc:38-50
    parts = ops.all_gather(rt, dev(rows[rank]))
    rt.sync()
    assert len(parts) == world and all(np.array_equal(p.cpu().numpy(), rows[r]) for r, p in enumerate(parts))
    done.append("abi_all_gather")
    # test_cuda_broadcast.cc:41-55: only the root holds the data
    x = dev([2.0, 3.0, 5.0, 6.0]) if rank == 0 else torch.zeros(4, device=f"cuda:{rank}")
    y = ops.broadcast(rt, x, 0)
    rt.sync()
    assert np.array_equal(y.cpu().numpy(), [2.0, 3.0, 5.0, 6.0])
    done.append("abi_broadcast")
    # test_cuda_sendrecv.cc:50-87: source 0 -> destination world-1 (2 in the reference's 3- and 4-rank cases)
    src, dst = 0, world - 1
    if world > 1 and rank == src:
        ops.send(rt, dev([2.0, 3.0, 5.0, 6.0]), dst)
    if world > 1 and rank == dst:
        got = ops.recv(rt, (2, 2), torch.float32, src)
        rt.sync()
        assert np.array_equal(got.cpu().numpy().ravel(), [2.0, 3.0, 5.0, 6.0])
    rt.sync()
    done.append("abi_send_recv")
    # an all-reduce captured in a hipGraph and replayed (collectives run on the runtime stream: capturable)
    buf = dev(rows[rank])
    out = torch.empty_like(buf)
    torch.cuda.synchronize()
    rt.use_own_stream()  # the legacy default stream cannot be captured
    ops.all_reduce(rt, "sum", buf, out=out)  # warm-up outside capture (RCCL sets up its channels on first use)
    rt.sync()
    out.zero_()
    torch.cuda.synchronize()
    rt.begin_capture()
    ops.all_reduce(rt, "sum", buf, out=out)
    g = rt.end_capture()
    for _ in range(3):
        rt.launch_graph(g)
    rt.sync()
    assert np.allclose(out.cpu().numpy(), want["sum"])
    done.append("abi_all_reduce_hipgraph")

    # ---- reference executor + plugin -----------------------------------------------------------------
    from conftest import load_backend_module

    B = load_backend_module()
    assert B is not None and hasattr(B, "RocmRuntime"), "plugin build missing"
    prt = B.RocmRuntime(rank)
    prt.init_comm("plugin_comm", world, rank)
    F32 = 1

    def graph(fn, data, shape=None):
        h = B.GraphHandler(prt)
        t = h.tensor(list(shape or [len(data)]), F32)
        out = fn(h, t)
        h.data_malloc()
        if data is not None:
            t.copyin_numpy(np.asarray(data, np.float32).reshape(shape or [len(data)]))
        h.run()
        return out

    for name, w in (("allReduceSum", want["sum"]), ("allReduceProd", want["prod"]), ("allReduceMin", want["min"]),
                    ("allReduceMax", want["max"]), ("allReduceAvg", want["avg"])):
        o = graph(lambda h, t: getattr(h, name)(t, None), rows[rank])
        assert np.allclose(o.copyout_numpy().ravel(), w, rtol=1e-6), (name, o.copyout_numpy(), w)
    outs = graph(lambda h, t: h.allGather(t, None, world), rows[rank])
    assert all(np.array_equal(o.copyout_numpy().ravel(), rows[r]) for r, o in enumerate(outs))
    o = graph(lambda h, t: h.broadcast(t, None, 0), [2.0, 3.0, 5.0, 6.0] if rank == 0 else [0.0] * 4)
    assert np.array_equal(o.copyout_numpy().ravel(), [2.0, 3.0, 5.0, 6.0])
    if world > 1 and rank == src:
        graph(lambda h, t: h.send(t, src, dst, None), [2.0, 3.0, 5.0, 6.0])
    if world > 1 and rank == dst:
        h = B.GraphHandler(prt)
        o = h.recv(None, src, dst, [2, 2], F32, None)
        h.data_malloc()
        h.run()
        assert np.array_equal(o.copyout_numpy().ravel(), [2.0, 3.0, 5.0, 6.0])
    done.append("plugin_collectives")
    # an AllReduceSum graph through run_with_hipgraph: captured once, replayed, same RCCL call count on every rank
    h = B.GraphHandler(prt)
    t = h.tensor([2], F32)
    o = h.allReduceSum(h.relu(t, None), None)
    h.data_malloc()
    for rep in range(3):
        t.copyin_numpy(np.asarray(rows[rank], np.float32) + rep)
        h.run_with_hipgraph()
        assert np.allclose(o.copyout_numpy().ravel(), want["sum"] + rep * world), rep
    assert prt.hip_graph_capture_count() == 1
    done.append("plugin_all_reduce_hipgraph")
    prt.sync()
    print("RESULT " + json.dumps({"rank": rank, "world": world, "done": done}), flush=True)


if __name__ == "__main__":
    main()
