"""Build the reference graph executor with the Device::ROCM plugin compiled in.

What is built:  infinitensor_amd/plugin/_build/backend.<abi>.so — the reference's own pybind module
`backend` (core + operators + utils + native-CPU kernels + ffi, compiled where they lie under
/root/reference) PLUS our plugin (infinitensor_amd/plugin/src/*.cc: RocmRuntimeObj and the
REGISTER_KERNEL'd Device::ROCM kernels), linked against infinitensor_amd/lib/libinfini_rocm.so.
Python then sees `backend.RocmRuntime(device)`, `.init_comm(name, world, rank)`,
`GraphHandler.run_with_hipgraph()` next to the untouched `GraphHandler` / `Tensor` API.

The reference needs five one-line touches to know a new device (SURVEY 8b "touch list"). They are NOT
stored in this repo as patched copies: this script applies them with exact-match string replacement
to temporary copies (a tmp dir that is deleted after the build) and fails if an anchor no longer
matches. INTEGRATION.md shows the same edits as a diff for a maintainer.

/root/reference does not exist on the GPU box: there the prebuilt .so (git-ignored, shipped by
gpurun) is used and this script is a no-op.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import sysconfig
import tempfile
from pathlib import Path

HERE = Path(__file__).resolve().parent
PKG = HERE.parent
REPO = PKG.parent
sys.path.insert(0, str(REPO))
from tools.cxxbuild import compile_all, link_shared  # noqa: E402

REF = Path("/root/reference")
OUT = HERE / "_build"

# (reference-relative file, anchor, replacement) — each anchor must match exactly once.
PATCHES = [
    ("include/core/runtime.h",
     "enum class Device { CPU = 1, CUDA, BANG, INTELCPU, KUNLUN, ASCEND };",
     "enum class Device { CPU = 1, CUDA, BANG, INTELCPU, KUNLUN, ASCEND, ROCM };"),
    ("include/core/runtime.h",
     "    bool isAscend() const { return device == Device::ASCEND; }",
     "    bool isAscend() const { return device == Device::ASCEND; }\n"
     "    bool isRocm() const { return device == Device::ROCM; }"),
    ("src/core/lazy_allocator.cc",
     "    if (runtime->isCuda()) {",
     "    if (runtime->isCuda() || runtime->isRocm()) { // 256-B arena alignment for 16-B vector access"),
    ("src/utils/operator_utils.cc",
     "    case Device::ASCEND:\n        return \"ASCEND\";",
     "    case Device::ASCEND:\n        return \"ASCEND\";\n    case Device::ROCM:\n        return \"ROCM\";"),
    ("include/core/graph_handler.h",
     "#ifdef USE_CUDA\n    inline void run_with_cudagraph() {",
     "#ifdef USE_ROCM\n    inline void run_with_hipgraph() {\n"
     "        (as<RocmRuntimeObj>(g->getRuntime()))->runWithHipGraph(g);\n    }\n"
     "    inline std::vector<std::string> rocm_fusion_plan() {\n"
     "        return RocmRuntimeObj::describeFusionPlan(g);\n    }\n#endif\n"
     "#ifdef USE_CUDA\n    inline void run_with_cudagraph() {"),
    ("include/core/graph_handler.h",
     "#include \"core/graph.h\"",
     "#include \"core/graph.h\"\n#ifdef USE_ROCM\n#include \"rocm/rocm_runtime.h\"\n#endif"),
    ("src/ffi/ffi_infinitensor.cc",
     "#ifdef USE_BANG\n#include \"bang/bang_runtime.h\"\n#endif",
     "#ifdef USE_BANG\n#include \"bang/bang_runtime.h\"\n#endif\n#ifdef USE_ROCM\n#include \"rocm/rocm_runtime.h\"\n#endif"),
    ("src/ffi/ffi_infinitensor.cc",
     "#ifdef USE_BANG\n    py::class_<BangRuntimeObj, std::shared_ptr<BangRuntimeObj>, RuntimeObj>(",
     "#ifdef USE_ROCM\n"
     "    py::class_<RocmRuntimeObj, std::shared_ptr<RocmRuntimeObj>, RuntimeObj>(\n"
     "        m, \"RocmRuntime\")\n"
     "        .def(py::init<int, size_t>(), py::arg(\"device\") = 0,\n"
     "             py::arg(\"hip_graph_cache_capacity\") = 16)\n"
     "        .def(\"clear_hip_graph_cache\", &RocmRuntimeObj::clearHipGraphCache)\n"
     "        .def(\"hip_graph_cache_size\", &RocmRuntimeObj::getHipGraphCacheSize)\n"
     "        .def(\"hip_graph_capture_count\", &RocmRuntimeObj::getHipGraphCaptureCount)\n"
     "        .def(\"sync\", &RocmRuntimeObj::sync)\n"
     "        .def(\"set_fusion\", &RocmRuntimeObj::setFusion)\n"
     "        .def(\"get_fusion\", &RocmRuntimeObj::getFusion)\n"
     "        .def(\"fused_launch_count\", &RocmRuntimeObj::getFusedLaunchCount)\n"
     "        .def(\"bridged_input_count\", &RocmRuntimeObj::getBridgedInputCount)\n"
     "        .def(\"parked_member_count\", &RocmRuntimeObj::getParkedMemberCount)\n"
     "        .def(\"forwarded_output_count\", &RocmRuntimeObj::getForwardedCount)\n"
     "        .def_static(\"save_perf\", &RocmRuntimeObj::savePerfData)\n"
     "        .def_static(\"load_perf\", &RocmRuntimeObj::loadPerfData)\n"
     "        .def_static(\"clear_perf\", &RocmRuntimeObj::clearPerfData)\n"
     "        .def_static(\"perf_size\", &RocmRuntimeObj::perfDataSize)\n"
     "        .def(\"init_comm\", &RocmRuntimeObj::initComm);\n"
     "#endif\n"
     "#ifdef USE_BANG\n    py::class_<BangRuntimeObj, std::shared_ptr<BangRuntimeObj>, RuntimeObj>("),
    ("src/ffi/ffi_infinitensor.cc",
     "        .def(\"run\", &Handler::run, policy::automatic)\n",
     "        .def(\"run\", &Handler::run, policy::automatic)\n"
     "#ifdef USE_ROCM\n        .def(\"run_with_hipgraph\", &Handler::run_with_hipgraph,\n"
     "             policy::automatic)\n"
     "        .def(\"rocm_fusion_plan\", &Handler::rocm_fusion_plan)\n#endif\n"),
]


def module_path() -> Path:
    return OUT / f"backend{sysconfig.get_config_var('EXT_SUFFIX')}"


def _apply_patches(overlay: Path) -> dict[str, Path]:
    patched: dict[str, str] = {}
    for rel, anchor, repl in PATCHES:
        text = patched.get(rel) or (REF / rel).read_text()
        if text.count(anchor) != 1:
            raise RuntimeError(f"patch anchor for {rel} matches {text.count(anchor)} times: {anchor[:60]!r}")
        patched[rel] = text.replace(anchor, repl)
    out = {}
    for rel, text in patched.items():
        dst = overlay / rel
        dst.parent.mkdir(parents=True, exist_ok=True)
        dst.write_text(text)
        out[rel] = dst
    return out


def build(verbose: bool = True) -> Path | None:
    out = module_path()
    if not REF.exists():
        if verbose:
            print(f"[plugin] {REF} absent; using prebuilt {out} ({'present' if out.exists() else 'MISSING'})")
        return out if out.exists() else None
    import pybind11

    lib = PKG / "lib" / "libinfini_rocm.so"
    assert lib.exists(), "build libinfini_rocm.so first (infinitensor_amd/build.py)"
    plugin_srcs = sorted((HERE / "src").glob("*.cc"))
    # the planner's rule families live in src/rocm_fusion_rules_*.inc, included inside class FusionPlanner (rocm_fusion.cc): an
    # edited .inc makes that one TU stale
    incs = sorted((HERE / "src").glob("*.inc"))
    fusion_cc = HERE / "src" / "rocm_fusion.cc"
    if incs and max(p.stat().st_mtime for p in incs) > fusion_cc.stat().st_mtime:
        os.utime(fusion_cc, None)
    newest_in = max(p.stat().st_mtime for p in [*plugin_srcs, *incs, HERE / "include/rocm/rocm_runtime.h", Path(__file__),
                                                REPO / "include/infini_rocm.h"])
    if out.exists() and out.stat().st_mtime >= newest_in:
        return out
    # stable path so that the compile flags (and with them the object-file names) do not change per build
    overlay = Path(tempfile.gettempdir()) / "irocm_plugin_overlay"
    shutil.rmtree(overlay, ignore_errors=True)
    overlay.mkdir(parents=True)
    try:
        patched = _apply_patches(overlay)
        inc = [
            f"-I{overlay}/include",           # patched headers shadow the reference's
            f"-I{HERE}/include",
            f"-I{REPO}/include",
            f"-I{REF}/include",
            f"-I{HERE}/third_party_shim",      # nlohmann/json.hpp shim (json 3.1.1) — the product build names nothing under oracle/
            f"-I{pybind11.get_include()}",
            f"-I{sysconfig.get_paths()['include']}",
        ]
        flags = ["-std=c++17", "-O2", "-fopenmp", "-fPIC", "-w", "-DUSE_ROCM=1", *inc]
        srcs = []
        for sub in ("core", "operators", "utils", "kernels/cpu", "ffi"):
            for f in sorted((REF / "src" / sub).glob("*.cc")):
                if f.name == "perf_engine.cc":
                    continue
                rel = str(f.relative_to(REF))
                srcs.append(patched.get(rel, f))
        # perf_engine.cc's replacement is plugin/src/rocm_perf.cc (JSON persistence written for json 3.1.1)
        srcs.extend(plugin_srcs)
        # every TU sees the patched runtime.h, so nothing can be shared with oracle/_ref's objects
        objdir = OUT / "obj"
        # a changed HEADER (or patch set) recompiles everything; a changed plugin source only itself (compile_all compares every
        # object with its own source's mtime)
        header_stamp = max(p.stat().st_mtime for p in [HERE / "include/rocm/rocm_runtime.h", Path(__file__), REPO / "include/infini_rocm.h"])
        objs = compile_all(srcs, objdir, flags, stamp=header_stamp)
        ldflags = subprocess.check_output(["python3-config", "--ldflags", "--embed"], text=True).split()
        link_shared(objs, out, ["-fopenmp", f"-L{lib.parent}", "-linfini_rocm", "-Wl,-rpath,$ORIGIN/../../lib", *ldflags])
    finally:
        shutil.rmtree(overlay, ignore_errors=True)
    return out


if __name__ == "__main__":
    print(build())
