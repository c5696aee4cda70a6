"""infinitensor_amd — MI355X-native (gfx950) operator backend for InfiniTensor.

Layout
  csrc/     hand-written HIP kernels + the C ABI (include/infini_rocm.h)
  lib/      built libinfini_rocm.so (git-ignored, shipped to the GPU box)
  plugin/   C++ `Device::ROCM` plugin for the reference graph executor
            (RocmRuntimeObj + REGISTER_KERNEL'd Kernel classes), built against /root/reference
  runtime.py / ops.py   ctypes host mirror of the reference Runtime / Kernel::compute glue

There is no CPU or PyTorch compute fallback: every op goes through the HIP library or raises.
"""
from ._lib import InfiniRocmError, declared_symbols, lib  # noqa: F401
from .runtime import DType, RocmRuntime  # noqa: F401

__all__ = ["InfiniRocmError", "RocmRuntime", "DType", "lib", "declared_symbols"]
