"""Extract the known-answer vectors of the reference's own kernel tests into tests/golden/kats.json.

Run in the build container (needs /root/reference):   python tests/golden/extract_kats.py
The GPU box has no /root/reference; tests there read the committed kats.json.

Every brace literal `vector<T>{...}`, `ExpectOutput{...}` or `Shape{...}` in the listed test
files is recorded with its file and 1-based line, so a test can cite e.g.
("test/kernels/cuda/test_cuda_matmul.cc", 50) and fetch exactly the numbers the reference
asserts at that line. Nothing else of the reference tests is copied: inputs, generators and
attributes are re-stated by our tests with the citation.
"""
from __future__ import annotations

import json
import re
import sys
from pathlib import Path

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent / "kats.json"

FILES = [
    "test/kernels/cuda/test_cuda_matmul.cc",
    "test/kernels/intelcpu/test_mkl_matmul.cc",
    "test/kernels/cuda/test_cuda_conv.cc",
    "test/kernels/intelcpu/test_mkl_conv.cc",
    "test/kernels/cuda/test_cuda_conv_fp16.cc",
    "test/kernels/cuda/test_cuda_softmax.cc",
    "test/kernels/intelcpu/test_mkl_softmax.cc",
    "test/kernels/cuda/test_cuda_layernorm.cc",
    "test/kernels/cuda/test_cuda_element_wise.cc",
    "test/kernels/intelcpu/test_mkl_element_wise.cc",
    "test/kernels/nativecpu/test_nativecpu_elementwise.cc",
    "test/kernels/cuda/test_cuda_unary.cc",
    "test/kernels/cuda/test_cuda_reduce.cc",
    "test/kernels/intelcpu/test_mkl_reduce.cc",
    "test/kernels/cuda/test_cuda_batch_norm.cc",
    "test/kernels/intelcpu/test_mkl_batch_norm.cc",
    "test/kernels/cuda/test_cuda_pooling.cc",
    "test/kernels/intelcpu/test_mkl_pooling.cc",
    "test/kernels/cuda/test_cuda_transpose.cc",
    "test/kernels/nativecpu/test_nativecpu_transpose.cc",
    "test/kernels/cuda/test_cuda_gather.cc",
    "test/kernels/cuda/test_cuda_where.cc",
    "test/kernels/cuda/test_cuda_concat.cc",
    "test/kernels/cuda/test_cuda_split.cc",
    "test/kernels/cuda/test_cuda_slice.cc",
    "test/kernels/cuda/test_cuda_pad.cc",
    "test/kernels/cuda/test_cuda_expand.cc",
    "test/kernels/cuda/test_cuda_reshape.cc",
    "test/kernels/cuda/test_cuda_clip.cc",
    "test/kernels/cuda/test_cuda_rope.cc",
    "test/kernels/cuda/test_cuda_attention.cc",
    "test/kernels/cuda/test_cuda_gather_elements.cc",
    "test/kernels/cuda/test_cuda_extend.cc",
    "test/kernels/cuda/test_cuda_resize.cc",
    "test/kernels/cuda/test_cuda_conv_transposed_2d.cc",
    "test/kernels/cuda/test_cuda_all_reduce.cc",
    "test/kernels/cuda/test_cuda_all_gather.cc",
    "test/kernels/cuda/test_cuda_broadcast.cc",
    "test/kernels/cuda/test_cuda_sendrecv.cc",
    "test/cuda/test_nccl_comm.cc",
]

LIT = re.compile(r"(vector\s*<\s*([\w:]+)\s*>|ExpectOutput|Shape)\s*\{([^{}]*)\}", re.S)
NUM = re.compile(r"[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?|inf|INFINITY)")


def parse_values(body: str):
    body = re.sub(r"//.*", "", body)
    toks = [t.strip() for t in body.replace("\n", " ").split(",") if t.strip()]
    vals = []
    for t in toks:
        t = t.rstrip("fFuUlL")
        if t in ("true", "false"):
            vals.append(1 if t == "true" else 0)
            continue
        m = NUM.fullmatch(t)
        if not m:
            return None  # not a pure numeric literal (e.g. expressions, identifiers)
        s = m.group(0)
        if re.fullmatch(r"[-+]?\d+", s):
            vals.append(int(s))
        else:
            vals.append(float(s))
    return vals


def main() -> int:
    if not REF.exists():
        print(f"{REF} not present: keeping committed {OUT.name}")
        return 0
    out = {}
    for rel in FILES:
        p = REF / rel
        if not p.exists():
            print("missing", rel)
            continue
        text = p.read_text()
        recs = []
        for m in LIT.finditer(text):
            vals = parse_values(m.group(3))
            if vals is None or len(vals) == 0:
                continue
            kind = "shape" if m.group(1) == "Shape" else (m.group(2) or "float")
            line = text.count("\n", 0, m.start()) + 1
            recs.append({"line": line, "kind": kind, "values": vals})
        out[rel] = recs
    OUT.write_text(json.dumps(out, indent=0, separators=(",", ":")))
    n = sum(len(v) for v in out.values())
    print(f"wrote {OUT} ({len(out)} files, {n} literals, {OUT.stat().st_size} bytes)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
