"""Build libinfini_rocm.so (the C-ABI library: HIP kernels + runtime + RCCL communicator) for gfx950.

`hipcc --offload-arch=gfx950` cross-compiles without a GPU. Output is in-tree
(infinitensor_amd/lib/libinfini_rocm.so) so that it ships to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
REPO = PKG.parent
sys.path.insert(0, str(REPO))
from tools.cxxbuild import compile_all, link_shared  # noqa: E402

CSRC = PKG / "csrc"
LIB = PKG / "lib" / "libinfini_rocm.so"
ROCM = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))


def build(verbose: bool = True) -> Path:
    srcs = sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.cc"))
    headers = list(CSRC.glob("*.h")) + [REPO / "include" / "infini_rocm.h"]
    stamp = max(h.stat().st_mtime for h in headers)
    flags = [
        "--offload-arch=gfx950",
        "-O3",
        "-std=c++17",
        "-fPIC",
        "-mcode-object-version=5",
        "-Wno-unused-result",
        f"-I{REPO / 'include'}",
        f"-I{CSRC}",
        f"-I{ROCM / 'include'}",
    ]
    objs = compile_all(srcs, PKG / "lib" / "obj", flags, compiler=str(ROCM / "bin" / "hipcc"), stamp=stamp)
    link_shared(
        objs,
        LIB,
        ["--offload-arch=gfx950", f"-L{ROCM / 'lib'}", "-lrccl", f"-Wl,-rpath,{ROCM / 'lib'}"],
        compiler=str(ROCM / "bin" / "hipcc"),
    )
    return LIB


if __name__ == "__main__":
    print(build())
