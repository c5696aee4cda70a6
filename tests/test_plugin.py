"""Device::ROCM plugin for the reference graph executor — checks that need no GPU: the build exists where
/root/reference is present, exposes the Python surface SURVEY 8b asks for, fails loudly without a device,
and leaves the reference's native-CPU path untouched."""
import numpy as np
import pytest
from conftest import REPO, kat


def test_plugin_build_is_present_where_reference_exists():
    import sysconfig
    from pathlib import Path

    p = REPO / "infinitensor_amd" / "plugin" / "_build" / f"backend{sysconfig.get_config_var('EXT_SUFFIX')}"
    if Path("/root/reference").exists():
        assert p.exists(), "run __graft_entry__.build()"


def test_python_surface(plugin_backend):
    b = plugin_backend
    assert hasattr(b, "RocmRuntime")
    for name in ("init_comm", "sync", "hip_graph_cache_size", "hip_graph_capture_count", "clear_hip_graph_cache"):
        assert hasattr(b.RocmRuntime, name), name
    assert hasattr(b.GraphHandler, "run_with_hipgraph")
    # everything else on GraphHandler / Tensor is the reference's (ffi_infinitensor.cc:478-637)
    for name in ("matmul", "conv", "softmax", "layerNormalization", "add", "relu", "reduceMean", "allReduceSum", "run", "tune"):
        assert hasattr(b.GraphHandler, name), name


def test_rocm_runtime_fails_loudly_without_device(plugin_backend):
    import torch

    if torch.cuda.is_available():
        pytest.skip("has a GPU")
    with pytest.raises(RuntimeError) as e:
        plugin_backend.RocmRuntime(0)
    assert "infini_rocm_runtime_create" in str(e.value)


def test_native_cpu_path_unchanged(plugin_backend):
    """The patched build still reproduces the reference KAT on its own CPU runtime (test_cuda_matmul.cc:63-65)."""
    b = plugin_backend
    h = b.GraphHandler(b.cpu_runtime())
    x, w = h.tensor([3, 5], 1), h.tensor([5, 2], 1)
    y = h.matmul(x, w, None, False, False, None, b.ActType.Linear, "default")
    h.data_malloc()
    x.copyin_float(list(map(float, range(15))))
    w.copyin_float(list(map(float, range(10))))
    h.run()
    assert np.array_equal(np.array(y.copyout_float()), kat("test/kernels/cuda/test_cuda_matmul.cc", 65, "float"))


def test_perf_engine_json_round_trip(plugin_backend, tmp_path):
    """The autotune cache (plugin/src/rocm_perf.cc): records in the reference's PerfEngine document layout
    (perf_engine.cc:23-45) — a ROCM MatMul variant record (type 3), a ROCM Conv variant record (type 4) and a plain
    timing record (type 0) — load, count, save, and come back identical; then records produced by the reference's own
    tune() on its CPU runtime survive a save / clear / load cycle."""
    import json

    R = plugin_backend.RocmRuntime
    R.clear_perf()
    assert R.perf_size() == 0
    ROCM = 7  # enum class Device { CPU = 1, ..., ROCM } (build_plugin.py patch of include/core/runtime.h:35)
    doc = {"data": [
        [[[ROCM, 101], {"hashType": 1234567890123456789, "opType": 101, "attrs": [101, 1, 4096, 4096, 4096, 0, 0, 0]}],
         {"type": 3, "data": [4, 0.104, "persist256"]}],
        [[[ROCM, 30], {"hashType": 42, "opType": 30, "attrs": [30, 128, 64, 56, 56, 64, 3, 3, 1, 1, 1, 1, 1, 1, 0]}],
         {"type": 4, "data": [2, 0.096, "conv_s1"]}],
        [[[1, 5], {"hashType": 7, "opType": 5, "attrs": [5, 2, 3]}], {"type": 0, "data": 3}],
    ]}
    src = tmp_path / "perf_in.json"
    src.write_text(json.dumps(doc))
    R.load_perf(str(src))
    assert R.perf_size() == 3
    dst = tmp_path / "perf_out.json"
    R.save_perf(str(dst))
    back = json.loads(dst.read_text())
    key = lambda e: (e[0][0][0], e[0][0][1], e[0][1]["hashType"])
    assert sorted(back["data"], key=key) == sorted(doc["data"], key=key)

    # Variant numbers are an implementation detail that has been renumbered between rounds: records are resolved by the
    # variant's NAME. A stale file — no name (round-2 schema: variant 7 no longer exists, variant 6 means something else
    # now), or a name this build does not know, or a number that disagrees with the name — never throws at launch: it
    # loads as "heuristic" (-1), or as the variant the name denotes.
    stale = {"data": [
        [[[ROCM, 101], {"hashType": 1, "opType": 101, "attrs": [101, 1, 64, 64, 64, 0, 0, 0]}], {"type": 3, "data": [7, 0.2]}],
        [[[ROCM, 101], {"hashType": 2, "opType": 101, "attrs": [101, 1, 64, 64, 64, 0, 0, 1]}], {"type": 3, "data": [6, 0.2, "stagger_st16"]}],
        [[[ROCM, 101], {"hashType": 3, "opType": 101, "attrs": [101, 1, 64, 64, 64, 0, 1, 0]}], {"type": 3, "data": [6, 0.2, "tile256_splitk"]}],
    ]}
    src.write_text(json.dumps(stale))
    R.load_perf(str(src))
    R.save_perf(str(dst))
    got = {e[0][1]["hashType"]: e[1]["data"] for e in json.loads(dst.read_text())["data"]}
    assert got == {1: [-1, 0.2, "heuristic"], 2: [-1, 0.2, "heuristic"], 3: [3, 0.2, "tile256_splitk"]}, got

    # records made by the reference's tune() itself
    R.clear_perf()
    b = plugin_backend
    h = b.GraphHandler(b.cpu_runtime())
    x, w = h.tensor([3, 5], 1), h.tensor([5, 2], 1)
    h.relu(h.matmul(x, w, None, False, False, None, b.ActType.Linear, "default"), None)
    h.data_malloc()
    x.copyin_float([1.0] * 15)
    w.copyin_float([1.0] * 10)
    h.tune()
    n = R.perf_size()
    assert n == 2
    R.save_perf(str(dst))
    R.clear_perf()
    assert R.perf_size() == 0
    R.load_perf(str(dst))
    assert R.perf_size() == n
    R.clear_perf()
    with pytest.raises(RuntimeError):
        R.load_perf(str(tmp_path / "does_not_exist.json"))
