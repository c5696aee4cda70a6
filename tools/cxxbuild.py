"""Tiny parallel C++ build helper (no cmake): compile a list of translation units
with a fixed flag set into an object directory, then link them.

Used by oracle/build_ref.py (reference native-CPU oracle) and by
infinitensor_amd/plugin/build_plugin.py (reference core + the ROCm plugin).
Incremental: a TU is recompiled only when its source (or any listed dependency
stamp) is newer than its object file.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys
from pathlib import Path
from typing import Iterable, Sequence


def _obj_name(src: Path, flags: Sequence[str] = ()) -> str:
    # path + flag set hashed into the name: changing flags recompiles
    h = hashlib.sha1((str(src) + "\0" + "\0".join(flags)).encode()).hexdigest()[:8]
    return f"{src.stem}.{h}.o"


def compile_all(
    sources: Iterable[Path],
    objdir: Path,
    flags: Sequence[str],
    compiler: str = "g++",
    jobs: int | None = None,
    stamp: float = 0.0,
) -> list[Path]:
    """Compile every source; return the object paths. Raises on first failure."""
    objdir.mkdir(parents=True, exist_ok=True)
    jobs = jobs or os.cpu_count() or 4
    work = []
    objs = []
    for src in sources:
        src = Path(src)
        obj = objdir / _obj_name(src, flags)
        objs.append(obj)
        if obj.exists() and obj.stat().st_mtime >= max(src.stat().st_mtime, stamp):
            continue
        work.append((src, obj))

    def one(item):
        src, obj = item
        cmd = [compiler, *flags, "-c", str(src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stderr

    if work:
        print(f"[cxxbuild] compiling {len(work)} TU(s) with {compiler} -j{jobs}", flush=True)
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        for src, rc, err in ex.map(one, work):
            if rc != 0:
                sys.stderr.write(err)
                raise RuntimeError(f"compile failed: {src}")
    return objs


def link_shared(objs: Sequence[Path], out: Path, flags: Sequence[str], compiler: str = "g++") -> None:
    out.parent.mkdir(parents=True, exist_ok=True)
    if out.exists() and all(out.stat().st_mtime >= o.stat().st_mtime for o in objs):
        return
    cmd = [compiler, "-shared", "-o", str(out), *map(str, objs), *flags]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr)
        raise RuntimeError(f"link failed: {out}")
    print(f"[cxxbuild] linked {out}", flush=True)
