"""Register / scratch budget of the SHIPPED device code (no GPU needed): the gfx950 code objects are extracted from
infinitensor_amd/lib/libinfini_rocm.so (llvm-objdump --offloading) and their kernel metadata read (llvm-readelf --notes:
.vgpr_count, .vgpr_spill_count, .private_segment_fixed_size = scratch bytes per lane).

Why it is a test: the 256-column GEMM builds live at the 256-VGPR wall of two waves per SIMD, and a scratch reload carries an
s_waitcnt vmcnt(0) that drains the LDS-DMA pipeline (round 2 measured +40 us on one launch for 400 bytes of scratch). Round 4
found that hipcc hoists the epilogue's lane arithmetic across the K loop and took every persistent GEMM build back under the wall
(DESIGN.md section 8); this pins it: a kernel of the hot-path families that starts to spill fails here, on the CPU, before any
benchmark is run."""
import shutil
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
LIB = REPO / "infinitensor_amd" / "lib" / "libinfini_rocm.so"
LLVM = Path("/opt/rocm/lib/llvm/bin")

# kernels allowed to use scratch, with the most they may use (bytes per lane): everything else must use none
ALLOWED_SCRATCH = {
    # local coefficient arrays indexed at run time (cubic / linear interpolation; off every hot path)
    "irocm::resize_interp_kernel": 144,
    # the causal prefill variants at the VGPR limit of their occupancy class (2 registers; not on a bench graph)
    "irocm::attention_kernel<irocm::F16Traits, 64, 2, true, 2>": 8,
    "irocm::attention_kernel<irocm::Bf16Traits, 128, 2, true, 1>": 12,
    "irocm::attention_kernel<irocm::F16Traits, 128, 2, true, 1>": 12,
    # 256-column conv + residual copy: 64 residual registers on top of 128 accumulators (36-48 bytes before round 4; bf16: none)
    "irocm::g256p::gemm256p_kernel<irocm::F16Traits, true, false, 4, false, 2>": 12,
    # one-shot 256^2 kernel with BOTH operands M/N-major (transposing reads for A and B; no graph of the configs uses it)
    "irocm::g256::gemm256_kernel<irocm::Bf16Traits, false, false, false>": 12,
    "irocm::g256::gemm256_kernel<irocm::F16Traits, false, false, false>": 12,
    # the four-wave GEMM at 512 registers, builds with two operands of the same kind: ONE loop-invariant register, stored in front of the
    # tile loop and reloaded once per tile at the top of the epilogue (never inside a K loop; the NN / TT builds have none)
    "irocm::g128w::gemm128w_kernel<irocm::Bf16Traits, true, true>": 8,
    "irocm::g128w::gemm128w_kernel<irocm::F16Traits, true, true>": 8,
    "irocm::g128w::gemm128w_kernel<irocm::Bf16Traits, false, false>": 8,
    "irocm::g128w::gemm128w_kernel<irocm::F16Traits, false, false>": 8,
    # diagnostics (probe.hip, tools/mfma_ceiling.py): 512-register upper-bound kernels, the spilled register lives outside their loops
    "mfma_wave128_kernel<irocm::Bf16Traits, 4>": 16,
    "mfma_wave128_kernel<irocm::F16Traits, 4>": 16,
    "mfma_wave128i_kernel": 8,
}


def _kernels(tmp_path):
    work = tmp_path / "co"
    work.mkdir()
    shutil.copy(LIB, work / LIB.name)
    subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", LIB.name], cwd=work, check=True, capture_output=True)
    out = []
    for co in sorted(work.glob("*gfx950*")):
        notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True, check=True).stdout
        cur = None
        for line in notes.splitlines():
            line = line.strip()
            if line.startswith(".name:"):
                cur = {"name": line.split(":", 1)[1].strip()}
                out.append(cur)
            elif cur is not None and line.split(":")[0] in (".private_segment_fixed_size", ".vgpr_count", ".vgpr_spill_count",
                                                            ".sgpr_spill_count", ".group_segment_fixed_size"):
                k, v = line.split(":", 1)
                cur[k] = int(v)
    names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in out), capture_output=True, text=True, check=True).stdout.splitlines()
    for k, n in zip(out, names):
        k["demangled"] = n
    return out


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not LIB.exists():
        pytest.skip("libinfini_rocm.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    if not (LLVM / "llvm-objdump").exists() or not (LLVM / "llvm-readelf").exists() or shutil.which("c++filt") is None:
        pytest.skip("llvm-objdump / llvm-readelf / c++filt not available")
    ks = _kernels(tmp_path_factory.mktemp("resources"))
    assert len(ks) > 500, "the extraction found too few kernels to mean anything"
    return ks


def test_no_kernel_outside_the_allow_list_uses_scratch(kernels):
    offenders = []
    for k in kernels:
        scratch = k.get(".private_segment_fixed_size", 0)
        if scratch == 0 and k.get(".vgpr_spill_count", 0) == 0:
            continue
        limit = max((v for n, v in ALLOWED_SCRATCH.items() if k["demangled"].startswith("void " + n)), default=None)
        if limit is None or scratch > limit:
            offenders.append((scratch, k.get(".vgpr_spill_count"), k[".vgpr_count"], k["demangled"][:160]))
    assert not offenders, "kernels with scratch beyond the allow list:\n" + "\n".join(map(str, offenders))


def test_persistent_gemm_builds_are_below_the_register_wall(kernels):
    """Every non-conv build of the persistent GEMM (4 layouts x 3 tile widths x 2 dtypes + the timeline builds): no scratch, no VGPR
    spill, at most 256 VGPRs; the headline kernel (bf16, NN, 256 columns) with head-room to spare."""
    gemm = [k for k in kernels if k["demangled"].startswith("void irocm::g256p::gemm256p_kernel<") and k["demangled"].rstrip(")").split(", ")[-1].startswith("0>")]
    assert len(gemm) >= 24, len(gemm)
    for k in gemm:
        assert k.get(".private_segment_fixed_size", 0) == 0 and k.get(".vgpr_spill_count", 0) == 0, k["demangled"]
        assert k[".vgpr_count"] <= 256, k["demangled"]
    head = [k for k in gemm if "Bf16Traits, true, false, 4, false, 0>" in k["demangled"]]
    assert len(head) == 1 and head[0][".vgpr_count"] <= 250, head


def test_three_wave_kernels_keep_their_occupancy(kernels):
    """The BERT attention variants (D = 64) rely on three waves per SIMD: <= 168 VGPRs."""
    att = [k for k in kernels if k["demangled"].startswith("void irocm::attention_kernel<") and ", 64, 2, false" in k["demangled"]]
    assert att
    for k in att:
        assert k[".vgpr_count"] <= 168, (k[".vgpr_count"], k["demangled"])
