import json
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_KATS = None


def kat(file: str, line: int, kind: str | None = None, index: int = 0) -> np.ndarray:
    """Golden vector asserted by the reference test `file` at `line` (tests/golden/kats.json,
    produced by tests/golden/extract_kats.py). `index` picks among several literals on a line."""
    global _KATS
    if _KATS is None:
        _KATS = json.loads((REPO / "tests" / "golden" / "kats.json").read_text())
    recs = [r for r in _KATS[file] if r["line"] == line and (kind is None or r["kind"] == kind)]
    if not recs:
        raise KeyError(f"no golden literal at {file}:{line} (kind={kind})")
    return np.array(recs[index]["values"])


def load_backend_module():
    """Load the reference's pybind module `backend`. Only ONE copy can live in a process (pybind11 type
    registry), so the build that contains the Device::ROCM plugin is preferred when it exists — it is the
    same reference sources (core, operators, native-CPU kernels, ffi) plus the plugin and the five-line
    device patch of infinitensor_amd/plugin/build_plugin.py; otherwise the pure oracle/_ref build."""
    import importlib.util
    import sysconfig

    if "backend" in sys.modules:
        return sys.modules["backend"]
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    for p in (REPO / "infinitensor_amd" / "plugin" / "_build" / f"backend{suffix}", REPO / "oracle" / "_ref" / f"backend{suffix}"):
        if p.exists():
            import torch  # noqa: F401  one HIP runtime per process: torch's must be mapped first (see _lib.py)

            spec = importlib.util.spec_from_file_location("backend", p)
            mod = importlib.util.module_from_spec(spec)
            sys.modules["backend"] = mod
            spec.loader.exec_module(mod)
            mod.__irocm_path__ = str(p)
            return mod
    return None


@pytest.fixture(scope="session")
def ref_backend():
    """The reference's own native-CPU backend (`backend.cpu_runtime()`), built from /root/reference."""
    mod = load_backend_module()
    if mod is None:
        pytest.skip("reference backend not built (run __graft_entry__.build() where /root/reference exists)")
    return mod


@pytest.fixture(scope="session")
def plugin_backend():
    """The reference graph executor WITH the Device::ROCM plugin (`backend.RocmRuntime`)."""
    mod = load_backend_module()
    if mod is None or not hasattr(mod, "RocmRuntime"):
        pytest.skip("plugin build (infinitensor_amd/plugin/_build) not present")
    return mod


@pytest.fixture(scope="session")
def rt():
    """A RocmRuntime on cuda:0 launching on torch's current stream. Fails loudly without the lib."""
    import torch

    from infinitensor_amd import RocmRuntime

    assert torch.cuda.is_available(), "gpu test needs a GPU"
    r = RocmRuntime(0)
    r.use_torch_stream()
    return r
