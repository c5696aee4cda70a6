"""Runtime part of the C ABI on a real MI355X: memory, streams, workspace, hipGraph capture/replay
(reference semantics: test/cuda/test_cudagraph.cc) and a single-rank RCCL communicator."""
import ctypes

import numpy as np
import pytest
import torch

from infinitensor_amd import InfiniRocmError, RocmRuntime, ops
from infinitensor_amd.runtime import Event
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu


def test_device_info(rt):
    info = rt.device_info()
    assert info["arch"].startswith("gfx950"), info
    assert info["compute_units"] >= 200 and info["wavefront_size"] == 64


def test_alloc_copy_roundtrip():
    r = RocmRuntime(0)
    x = np.arange(1000, dtype=np.float32)
    p = r.alloc(x.nbytes)
    q = r.alloc(x.nbytes)
    r.copy_from_cpu(p, x.ctypes.data_as(ctypes.c_void_p), x.nbytes)
    r.copy_inside(q, p, x.nbytes)
    y = np.empty_like(x)
    r.copy_to_cpu(y.ctypes.data_as(ctypes.c_void_p), q, x.nbytes)
    assert np.array_equal(x, y)
    r.dealloc(p)
    r.dealloc(q)


def test_workspace_grows_and_is_stable(rt):
    a = rt.workspace(1 << 20)
    b = rt.workspace(1 << 10)
    assert a == b and a != 0
    c = rt.workspace(64 << 20)
    assert c != 0


def test_own_stream_and_events():
    r = RocmRuntime(0)  # own non-blocking stream
    x = torch.randn(1 << 20, device="cuda")
    torch.cuda.synchronize()
    e0, e1 = Event(), Event()
    r.record(e0)
    y = torch.empty_like(x)
    ops.unary(r, "relu", x, out=y)
    r.record(e1)
    assert r.elapsed_ms(e0, e1) >= 0
    r.sync()
    assert torch.equal(y, torch.clamp(x, min=0))  # torch used only to CHECK here


def test_graph_capture_and_replay_matches_eager():
    """Capture MatMul -> Add(bias) -> Gelu -> LayerNorm once, replay with new input contents in the same
    buffers (test_cudagraph.cc: capture-once-replay)."""
    r = RocmRuntime(0)
    rng = np.random.default_rng(0)
    a = torch.from_numpy(rng.standard_normal((256, 128)).astype(np.float32)).cuda()
    w = torch.from_numpy(rng.standard_normal((128, 192)).astype(np.float32)).cuda()
    b = torch.from_numpy(rng.standard_normal((192,)).astype(np.float32)).cuda()
    g = torch.from_numpy(rng.standard_normal((192,)).astype(np.float32)).cuda()
    t1, t2, t3, out = (torch.empty(256, 192, device="cuda") for _ in range(4))
    torch.cuda.synchronize()

    def chain():
        ops.matmul(r, a, w, out=t1)
        ops.binary(r, "add", t1, b, out=t2)
        ops.unary(r, "gelu", t2, out=t3)
        ops.layer_norm(r, t3, g, None, 1e-5, -1, out=out)

    chain()
    r.sync()
    eager = out.clone()
    r.begin_capture()
    chain()
    graph = r.end_capture()
    out.zero_()
    torch.cuda.synchronize()
    r.launch_graph(graph)
    r.sync()
    assert torch.equal(out, eager)
    a.copy_(torch.from_numpy(rng.standard_normal((256, 128)).astype(np.float32)))
    torch.cuda.synchronize()
    r.launch_graph(graph)
    r.sync()
    want = R.layer_norm(R.unary("gelu", R.matmul(a.cpu().numpy(), w.cpu().numpy(), b.cpu().numpy())), g.cpu().numpy(), None, 1e-5)
    assert np.allclose(out.cpu().numpy(), want, rtol=1e-4, atol=1e-4)


def test_capture_errors_are_reported():
    r = RocmRuntime(0)
    with pytest.raises(InfiniRocmError):
        r.end_capture()  # no capture active
    r.begin_capture()
    with pytest.raises(InfiniRocmError):
        r.begin_capture()
    r.abort_capture()
    x = torch.ones(16, device="cuda")
    torch.cuda.synchronize()
    y = ops.unary(r, "neg", x)  # the runtime is usable again after an aborted capture
    r.sync()
    assert float(y.sum()) == -16


def test_workspace_grows_inside_a_capture_and_retires_old_blocks():
    """Growing the scratch block is legal while the stream records (hipMalloc under a thread-local Relaxed capture
    mode) and never frees the outgrown block: launches recorded earlier in the same capture, and graph execs captured
    before, keep addressing live memory (csrc/runtime.hip; the reference's fixed 7 GiB block never moves either)."""
    r = RocmRuntime(0)
    p0 = r.workspace(1 << 20)
    info0 = r.workspace_info()
    assert info0["bytes"] >= 1 << 20 and info0["retired_blocks"] == 0
    x = torch.arange(1024, device="cuda", dtype=torch.float32)
    y = torch.zeros_like(x)
    torch.cuda.synchronize()
    r.begin_capture()
    r.copy_inside(p0, x.data_ptr(), 4096)            # recorded against the first block
    p1 = r.workspace(64 << 20)                        # grows inside the capture
    r.copy_inside(y.data_ptr(), p0, 4096)            # still reads the (now retired) first block
    g = r.end_capture()
    info1 = r.workspace_info()
    assert p1 != p0 and info1["retired_blocks"] == 1 and info1["epoch"] == info0["epoch"] + 1 and info1["bytes"] >= 64 << 20
    for _ in range(3):
        y.zero_()
        torch.cuda.synchronize()
        r.launch_graph(g)
        r.sync()
        assert torch.equal(y, x)
    assert r.workspace(1 << 20) == p1  # no shrink
    del g
    r.workspace_trim()
    assert r.workspace_info()["retired_blocks"] == 0


def test_single_rank_communicator(tmp_path, monkeypatch):
    """world_size = 1 RCCL communicator: all-reduce / all-gather / broadcast are identities
    (test_cuda_all_reduce.cc needs >= 2 GPUs; the 1-GPU box can only exercise the plumbing)."""
    monkeypatch.chdir(tmp_path)
    r = RocmRuntime(0)
    r.init_comm("test_comm", 1, 0)
    assert r.comm_info() == (1, 0)
    x = torch.arange(1024, device="cuda", dtype=torch.float32)
    torch.cuda.synchronize()
    for kind in ("sum", "prod", "min", "max", "avg"):
        y = ops.all_reduce(r, kind, x)
        r.sync()
        assert torch.equal(y, x), kind
    h = ops.all_reduce(r, "sum", x.half())
    (gathered,) = ops.all_gather(r, x)
    bc = ops.broadcast(r, x, 0)
    r.sync()
    assert torch.equal(h, x.half()) and torch.equal(gathered, x) and torch.equal(bc, x)
    with pytest.raises(InfiniRocmError):
        ops.send(r, x, 0)  # peer must differ from own rank


def test_collective_without_communicator_fails_loudly():
    r = RocmRuntime(0)
    with pytest.raises(InfiniRocmError, match="communicator not initialised"):
        ops.all_reduce(r, "sum", torch.ones(4, device="cuda"))


@pytest.mark.gpu
def test_wait_torch_orders_a_private_stream_behind_torch():
    """A runtime with a stream of its own does not order with torch's stream by itself; `wait_torch()` makes it (an event wait, no
    host sync). torch fills a large tensor (a long kernel on torch's stream); the runtime's own stream then copies it with a unary
    identity-like op — with the wait the copy always sees the filled values."""
    import torch

    from infinitensor_amd import RocmRuntime, ops

    rt = RocmRuntime(0)  # own non-blocking stream
    assert rt.stream() != torch.cuda.current_stream().cuda_stream
    for i in range(8):
        x = torch.empty(64 << 20, device="cuda", dtype=torch.float16)
        x.fill_(float(i + 1))
        x.mul_(2.0)
        rt.wait_torch()
        y = ops.unary(rt, "relu", x)
        rt.sync()
        assert torch.all(y == float(2 * (i + 1))).item(), i
    rt.use_torch_stream()
    rt.wait_torch()  # same stream: nothing to do
