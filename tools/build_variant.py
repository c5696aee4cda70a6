#!/usr/bin/env python3
"""Build a VARIANT of libinfini_rocm.so for same-box A/B runs: the named translation units recompiled with extra -D flags, every other
object taken from the regular build (run `python __graft_entry__.py` first), linked to infinitensor_amd/lib/ab/<name>.so. Select it
with INFINI_ROCM_LIB=<path> (infinitensor_amd/_lib.py).
usage: tools/build_variant.py NAME --tus gemm256p_nt4.hip[,more.hip] -DIROCM_KV=2 [-D...]"""
import argparse
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from tools.cxxbuild import compile_all, link_shared  # noqa: E402

PKG = REPO / "infinitensor_amd"
CSRC = PKG / "csrc"
ROCM = Path("/opt/rocm")


def build_variant(name: str, tus: list[str], defines: list[str]) -> Path:
    base_flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mcode-object-version=5", "-Wno-unused-result",
                  f"-I{REPO / 'include'}", f"-I{CSRC}", f"-I{ROCM / 'include'}"]
    srcs = sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.cc"))
    headers = list(CSRC.glob("*.h")) + [REPO / "include" / "infini_rocm.h"]
    stamp = max(h.stat().st_mtime for h in headers)
    hipcc = str(ROCM / "bin" / "hipcc")
    special = [s for s in srcs if s.name in tus]
    assert len(special) == len(tus), f"unknown translation unit in {tus}"
    rest = [s for s in srcs if s.name not in tus]
    objs = compile_all(rest, PKG / "lib" / "obj", base_flags, compiler=hipcc, stamp=stamp)
    objs += compile_all(special, PKG / "lib" / "obj_ab", base_flags + defines, compiler=hipcc, stamp=stamp)
    out = PKG / "lib" / "ab" / f"{name}.so"
    out.parent.mkdir(parents=True, exist_ok=True)
    link_shared(objs, out, ["--offload-arch=gfx950", f"-L{ROCM / 'lib'}", "-lrccl", f"-Wl,-rpath,{ROCM / 'lib'}"], compiler=hipcc)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--tus", default="gemm256p_nt4.hip")
    a, defs = ap.parse_known_args()
    print(build_variant(a.name, a.tus.split(","), defs))
