"""GPU probe: device facts + GEMM variant sweep (TFLOP/s by shape/layout/dtype), events on the runtime stream."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from infinitensor_amd import RocmRuntime, ops
from infinitensor_amd.runtime import Event


def main():
    rt = RocmRuntime(0)
    print(rt.device_info())
    variants = ops.matmul_variants()
    print("variants:", variants)
    shapes = [(1, 4096, 4096, 4096), (1, 8192, 8192, 8192), (1, 2048, 2048, 2048), (1, 16384, 768, 768),
              (1, 16384, 3072, 768), (1, 16384, 768, 3072), (384, 512, 512, 64), (384, 512, 64, 512),
              (1, 2048, 12288, 4096), (1, 2048, 4096, 11008), (1, 2048, 4096, 4096), (1, 2048, 512, 4096),
              (1, 2048, 4096, 1376), (1, 128, 1000, 2048)]
    import os
    if os.environ.get("PROBE_SHAPES") == "tp":
        shapes = shapes[-6:]
    for dt in (torch.bfloat16, torch.float16):
        for (b, m, n, k) in shapes:
            for ta, tb in ((False, False), (False, True)):
                if dt == torch.float16 and (m, n, k) != (4096, 4096, 4096):
                    continue
                a = torch.randn((b, k, m) if ta else (b, m, k), device="cuda").to(dt)
                w = torch.randn((b, n, k) if tb else (b, k, n), device="cuda").to(dt)
                c = torch.empty(b, m, n, device="cuda", dtype=dt)
                torch.cuda.synchronize()
                row = []
                for v in range(1, len(variants)):
                    ops.set_matmul_variant(rt, v)
                    for _ in range(3):
                        ops.matmul(rt, a, w, None, ta, tb, out=c)
                    iters = 20
                    e0, e1 = Event(), Event()
                    rt.record(e0)
                    for _ in range(iters):
                        ops.matmul(rt, a, w, None, ta, tb, out=c)
                    rt.record(e1)
                    ms = rt.elapsed_ms(e0, e1) / iters
                    row.append(f"{variants[v][-9:]}={2.0 * b * m * n * k / ms / 1e9:7.1f}TF")
                ops.set_matmul_variant(rt, -1)
                for _ in range(3):
                    ops.matmul(rt, a, w, None, ta, tb, out=c)
                e0, e1 = Event(), Event()
                rt.record(e0)
                for _ in range(20):
                    ops.matmul(rt, a, w, None, ta, tb, out=c)
                rt.record(e1)
                ms = rt.elapsed_ms(e0, e1) / 20
                row.append(f"heuristic={2.0 * b * m * n * k / ms / 1e9:7.1f}TF")
                ops.set_matmul_variant(rt, -1)
                print(f"{str(dt)[6:]:9s} b{b} m{m} n{n} k{k} tA{int(ta)} tB{int(tb)}: " + "  ".join(row), flush=True)
    # fp32 generic
    for (m, n, k) in [(512, 512, 512), (4096, 4096, 4096)]:
        a = torch.randn(m, k, device="cuda")
        w = torch.randn(k, n, device="cuda")
        c = torch.empty(m, n, device="cuda")
        torch.cuda.synchronize()
        for _ in range(2):
            ops.matmul(rt, a, w, out=c)
        e0, e1 = Event(), Event()
        rt.record(e0)
        for _ in range(5):
            ops.matmul(rt, a, w, out=c)
        rt.record(e1)
        ms = rt.elapsed_ms(e0, e1) / 5
        print(f"f32 m{m} n{n} k{k}: {2.0 * m * n * k / ms / 1e9:8.1f} TF ({ms * 1e3:.1f} us)")


if __name__ == "__main__":
    main()
