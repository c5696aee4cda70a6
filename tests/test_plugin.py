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
