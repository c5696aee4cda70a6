"""Counter evidence of THIS round must describe THIS tree (round-5 verdict, weak #3: the memory-bound counter file was taken four commits
before the last change to the kernels it measured, bench.py refused it and the driver's line carried no traffic figure). Every
profiles/r06_*_pmc.json that carries a source stamp is checked against the sources as they are now: touching csrc/ after the evidence
visit fails the CPU suite until tools/gpu_evidence.sh has run again."""
import json
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "tools"))
import source_stamps  # noqa: E402

ROUND = "r06"


def test_current_round_counter_files_match_the_sources():
    want = {"membound": source_stamps.membound_stamp(), "gemm": source_stamps.gemm_stamp()}
    stale, seen = [], 0
    for f in sorted((REPO / "profiles").glob(f"{ROUND}_*_pmc.json")):
        d = json.loads(f.read_text())
        if "stamp" not in d or d["stamp"] is None:
            continue
        seen += 1
        kind = "membound" if "membound" in f.name else ("gemm" if "_gemm" in f.name else None)
        if kind and d["stamp"] != want[kind]:
            stale.append(f"{f.name}: stamp {d['stamp']}, sources are {want[kind]}")
    assert not stale, "stale counter evidence (rerun tools/gpu_evidence.sh and copy its files): " + "; ".join(stale)


def test_bench_reads_this_rounds_counter_files_first():
    src = (REPO / "bench.py").read_text()
    assert f'"{ROUND}_membound_pmc.json"' in src and f'"{ROUND}_gemm_pmc.json"' in src
