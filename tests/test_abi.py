"""The C-ABI library loads and exports every symbol include/infini_rocm.h declares (no GPU)."""
import ctypes

import pytest
from conftest import REPO

from infinitensor_amd import _lib


def test_library_exists_and_loads():
    assert _lib.LIB_PATH.exists(), "libinfini_rocm.so not built"
    L = _lib.lib()
    assert b"gfx950" in L.infini_rocm_version()


def test_every_declared_symbol_is_exported():
    L = _lib.lib()
    names = _lib.declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in include/infini_rocm.h but not exported: {missing}"


def test_device_count_without_gpu_is_not_an_error():
    n = ctypes.c_int(-1)
    assert _lib.lib().infini_rocm_device_count(ctypes.byref(n)) == 0
    assert n.value >= 0


def test_runtime_create_fails_loudly_without_device():
    import torch

    if torch.cuda.is_available():
        return
    h = ctypes.c_void_p()
    st = _lib.lib().infini_rocm_runtime_create(0, ctypes.byref(h))
    assert st != 0
    assert _lib.lib().infini_rocm_last_error()


def test_no_oracle_import_in_product():
    """The product package must never import the oracle (tier rule 3)."""
    import re
    from pathlib import Path

    pkg = Path(_lib.__file__).parent
    for f in pkg.rglob("*.py"):
        txt = f.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f


def test_bench_has_no_cpu_fallback():
    """bench.py on a box without a GPU must fail loudly: non-zero exit, no JSON line (the product path never routes
    through the oracle or any CPU implementation)."""
    import subprocess
    import sys

    import torch

    if torch.cuda.is_available():
        pytest.skip("has a GPU")
    p = subprocess.run([sys.executable, str(REPO / "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300)
    assert p.returncode != 0
    assert not any(line.startswith("{") for line in p.stdout.splitlines())
    assert "GPU" in p.stderr
