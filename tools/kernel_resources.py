#!/usr/bin/env python3
"""Print VGPR / SGPR / scratch / LDS of the kernels in an object or shared library (gfx950 code objects, no GPU needed).
usage: tools/kernel_resources.py [file] [substring ...]   (default file: infinitensor_amd/lib/libinfini_rocm.so)"""
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")


def kernels(path: Path):
    out = []
    with tempfile.TemporaryDirectory() as td:
        work = Path(td)
        shutil.copy(path, work / path.name)
        subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", path.name], cwd=work, check=True, capture_output=True)
        for co in sorted(work.glob("*gfx950*")):
            notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True, check=True).stdout
            cur = None
            for line in notes.splitlines():
                line = line.strip()
                if line.startswith(".name:"):
                    cur = {"name": line.split(":", 1)[1].strip()}
                    out.append(cur)
                elif cur is not None and ":" in line and line.split(":")[0] in (
                    ".private_segment_fixed_size", ".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".group_segment_fixed_size"):
                    k, v = line.split(":", 1)
                    cur[k] = int(v)
    names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in out), capture_output=True, text=True, check=True).stdout.splitlines()
    for k, n in zip(out, names):
        k["demangled"] = n
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    path = Path(args[0]) if args and Path(args[0]).exists() else Path(__file__).resolve().parent.parent / "infinitensor_amd/lib/libinfini_rocm.so"
    subs = [a for a in args if not Path(a).exists()]
    print("vgpr agpr sgpr scratch vspill lds  kernel")
    for k in kernels(path):
        if subs and not any(s in k["demangled"] for s in subs):
            continue
        print(f'{k.get(".vgpr_count", 0):4d} {k.get(".agpr_count", 0):4d} {k.get(".sgpr_count", 0):4d} {k.get(".private_segment_fixed_size", 0):7d} '
              f'{k.get(".vgpr_spill_count", 0):6d} {k.get(".group_segment_fixed_size", 0):6d}  {k["demangled"][:150]}')
