"""Build the reference's own native-CPU backend into oracle/_ref/ (TEST INFRASTRUCTURE ONLY).

What is compiled: /root/reference/src/{core,operators,utils,kernels/cpu,ffi}/*.cc, where they
lie (never copied), with g++ -std=c++17, pip pybind11 and the json shim in oracle/shim/.
The only substitution is src/core/perf_engine.cc -> oracle/shim/perf_engine_stub.cc (json 3.1.1
lacks get_to). Output: oracle/_ref/backend.<abi>.so — the reference pybind module `backend`
(`backend.cpu_runtime()`, `backend.GraphHandler`), used to (a) validate oracle/ref_ops.py,
(b) generate tests/golden/*.npz, (c) serve as bench.py's cpu_baseline {"kind": "reference"}.

/root/reference does not exist on the GPU box: there the prebuilt .so (git-ignored, but shipped
by gpurun) is used as-is and this script is a no-op.
"""
from __future__ import annotations

import subprocess
import sys
import sysconfig
from pathlib import Path

HERE = Path(__file__).resolve().parent
REPO = HERE.parent
sys.path.insert(0, str(REPO))
from tools.cxxbuild import compile_all, link_shared  # noqa: E402

REF = Path("/root/reference")
OUT = HERE / "_ref"


def ext_suffix() -> str:
    return sysconfig.get_config_var("EXT_SUFFIX")


def ref_module_path() -> Path:
    return OUT / f"backend{ext_suffix()}"


def build(verbose: bool = True) -> Path | None:
    out = ref_module_path()
    if not REF.exists():
        if verbose:
            print(f"[oracle] {REF} absent; using prebuilt {out} ({'present' if out.exists() else 'MISSING'})")
        return out if out.exists() else None
    import pybind11

    inc = [
        f"-I{REF}/include",
        f"-I{HERE}/shim",
        f"-I{pybind11.get_include()}",
        f"-I{sysconfig.get_paths()['include']}",
    ]
    flags = ["-std=c++17", "-O2", "-fopenmp", "-fPIC", "-w", *inc]
    srcs = []
    for sub in ("core", "operators", "utils", "kernels/cpu", "ffi"):
        for f in sorted((REF / "src" / sub).glob("*.cc")):
            if f.name == "perf_engine.cc":
                continue
            srcs.append(f)
    srcs.append(HERE / "shim" / "perf_engine_stub.cc")
    objs = compile_all(srcs, OUT / "obj", flags)
    ldflags = subprocess.check_output(["python3-config", "--ldflags", "--embed"], text=True).split()
    link_shared(objs, out, ["-fopenmp", *ldflags])
    return out


if __name__ == "__main__":
    p = build()
    print(p)
