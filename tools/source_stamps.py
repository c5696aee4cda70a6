"""sha1 stamps of the kernel sources a counter file under profiles/ was taken from. A counter file is evidence for the kernels it measured
and for nothing else: tools/profile_membound.sh / tools/profile_gemm.sh write the stamp into their summaries, bench.py refuses a file whose
stamp differs from the sources it runs, and tests/test_profile_stamps_cpu.py FAILS the CPU suite while a current-round counter file is stale —
the evidence visit (tools/gpu_evidence.sh) has to be the last thing that touches csrc/."""
import hashlib
from pathlib import Path

CSRC = Path(__file__).resolve().parent.parent / "infinitensor_amd" / "csrc"
MEMBOUND_SOURCES = ("rowops.hip", "elementwise.hip", "movement.hip", "nnops.hip", "rope.hip", "common.h")
GEMM_SOURCES = ("gemm128w.hip", "gemm256p_kernel.h", "gemm256_common.h", "gemm_common.h", "gemm256p_nt4.hip", "common.h")


def _stamp(names) -> str:
    h = hashlib.sha1()
    for n in sorted(names):
        h.update((CSRC / n).read_bytes())
    return h.hexdigest()[:16]


def membound_stamp() -> str:
    return _stamp(MEMBOUND_SOURCES)


def gemm_stamp() -> str:
    return _stamp(GEMM_SOURCES)


if __name__ == "__main__":
    print("membound", membound_stamp(), "gemm", gemm_stamp())
