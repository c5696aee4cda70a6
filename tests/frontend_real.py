"""Load the reference's REAL front-end — `pyinfinitensor/src/pyinfinitensor/onnx.py`, unmodified, where it lies — for the tests.

The file is never copied into this repository: it is found in an installed `pyinfinitensor`, under $INFINITENSOR_PY_SRC, or under
/root/reference (the build container; the GPU boxes have none of them, the tests that need it skip there and the committed golden
graph signatures — tests/golden/onnx/*_frontend.json, written by tests/golden/make_frontend_goldens.py from a run of THIS loader — stand
in). It imports `onnx` / `onnxsim`; when the real packages are absent (this image) tests/onnx_shim supplies protobuf-backed stand-ins.
Its `import backend` gets the module tests/conftest.py loaded (the plugin build when present)."""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent


def ensure_onnx() -> str:
    """-> "real" or "shim": make `import onnx` / `import onnxsim` work."""
    try:
        import onnx  # noqa: F401
        import onnxsim  # noqa: F401

        return "shim" if getattr(sys.modules["onnx"], "IS_SHIM", False) else "real"
    except ImportError:
        for m in ("onnx", "onnxsim"):
            sys.modules.pop(m, None)
        sys.path.insert(0, str(HERE / "onnx_shim"))
        import onnx  # noqa: F401
        import onnxsim  # noqa: F401

        return "shim"


def frontend_path() -> Path | None:
    for cand in (os.environ.get("INFINITENSOR_PY_SRC"), "/root/reference/pyinfinitensor/src"):
        if cand and (Path(cand) / "pyinfinitensor" / "onnx.py").exists():
            return Path(cand) / "pyinfinitensor" / "onnx.py"
    try:
        spec = importlib.util.find_spec("pyinfinitensor")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.origin:
        p = Path(spec.origin).parent / "onnx.py"
        if p.exists():
            return p
    return None


_MOD = None


def load_frontend(backend_module):
    """The reference's onnx.py as a module (None when it is nowhere to be found). `backend_module` must already be sys.modules["backend"]."""
    global _MOD
    if _MOD is not None:
        return _MOD
    path = frontend_path()
    if path is None:
        return None
    assert sys.modules.get("backend") is backend_module, "onnx.py does `import backend`: it must get the module the tests loaded"
    ensure_onnx()
    spec = importlib.util.spec_from_file_location("pyinfinitensor_onnx_real", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _MOD = mod
    return mod
