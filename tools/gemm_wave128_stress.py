"""Race hunt for the four-wave GEMM (csrc/gemm128w.hip: counted waits, hand-kept hazards): the same inputs must give the SAME BITS launch
after launch, in every layout, at sizes from one tile to several per workgroup, with other work interleaved; and agree with the eight-wave
kernel within the parity bound.  python tools/gemm_wave128_stress.py [rounds]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from infinitensor_amd import RocmRuntime, ops

rt = RocmRuntime(0)
rt.use_torch_stream()
w = ops.matmul_variants().index("wave128")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
bad = 0
g = torch.Generator(device="cuda").manual_seed(77)
for r in range(rounds):
    for (bt, m, n, k) in [(1, 256, 256, 128), (1, 4096, 4096, 4096), (1, 8192, 4096, 2048), (3, 1024, 768, 640), (1, 5120, 3328, 1152), (2, 2048, 2048, 3968)]:
        for ta in (False, True):
            for tb in (False, True):
                dt = torch.bfloat16 if (r + ta + tb) % 2 == 0 else torch.float16
                a = torch.randn((bt, k, m) if ta else (bt, m, k), device="cuda", generator=g).to(dt)
                b = torch.randn((bt, n, k) if tb else (bt, k, n), device="cuda", generator=g).to(dt)
                ops.set_matmul_variant(rt, w)
                outs = []
                for rep in range(4):
                    outs.append(ops.matmul(rt, a, b, None, ta, tb))
                    if rep == 1:  # something else on the stream in between: LDS / cache state differs
                        ops.softmax(rt, a.reshape(-1, a.shape[-1])[:512], axis=1)
                assert ops.matmul_last_variant(rt) == "wave128"
                ops.set_matmul_variant(rt, 4)
                ref = ops.matmul(rt, a, b, None, ta, tb)
                ops.set_matmul_variant(rt, -1)
                torch.cuda.synchronize()
                same = all(torch.equal(outs[0], o) for o in outs[1:])
                tol = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
                a64 = (a.double().transpose(-1, -2) if ta else a.double())
                b64 = (b.double().transpose(-1, -2) if tb else b.double())
                slack = tol * ref.double().abs() + 2.0 ** -16 * (a64.abs() @ b64.abs())
                close = bool(((outs[0].double() - ref.double()).abs() <= 2 * slack).all())
                if not (same and close):
                    bad += 1
                    print(f"FAIL round {r} {dt} b{bt} m{m} n{n} k{k} ta{int(ta)} tb{int(tb)}: deterministic {same}, agrees with persist256 {close}", flush=True)
print(f"{rounds} rounds, {bad} failures")
sys.exit(1 if bad else 0)
