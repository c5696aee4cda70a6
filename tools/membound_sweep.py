"""HBM-roofline sweep of the memory-bound operators (round-3 verdict, "do this" #4): every row at the shape the bench
graphs run it (configs 3 / 4 / 5) AND at an HBM-sized shape (>= 512 MB of algorithmic traffic, beyond the 256 MiB Infinity
Cache), timed with HIP events on the runtime stream after a time-based warm-up.

  python tools/membound_sweep.py [--only rope,gather] [--json out.json]
Algorithmic bytes = every input element read once + every output element written once (a gathered table counts its
gathered rows, a broadcast operand its own size). `frac` is against the 8 TB/s HBM3E peak of MI355X_MICROARCH.md (the guide
measures 6.29 TB/s = 0.79 as the achievable copy rate).
Used by bench.py (`extras.membound`) and by tools/profile_membound.sh for the PMC traffic figures."""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
PEAK_HBM_GBS = 8000.0


def timeit(rt, Event, fn, min_ms=20.0):
    """Seconds per call: the better of two timed passes (one pass of the gather row once read 275 us against 111-115 in every other run)."""
    return min(_timeit_once(rt, Event, fn, min_ms), _timeit_once(rt, Event, fn, min_ms / 2))


def _timeit_once(rt, Event, fn, min_ms=20.0):
    fn()
    rt.sync()
    e0, e1 = Event(), Event()
    rt.record(e0)
    for _ in range(3):
        fn()
    rt.record(e1)
    per = max(rt.elapsed_ms(e0, e1) / 3, 1e-3)
    if per < 0.05:
        # A Python call costs 10-20 us: a kernel shorter than that is timed as a hipGraph of 20 calls, replayed (round 5: the
        # config-shape Split / Concat / ReduceMean rows were host-bound, not kernel-bound). Runtimes on a stream that cannot be
        # captured (torch's legacy default stream) keep the call loop.
        try:
            rt.sync()
            rt.begin_capture()
            try:
                for _ in range(20):
                    fn()
            except Exception:
                rt.abort_capture()
                raise
            g = rt.end_capture()
            for _ in range(int(25.0 / (per * 20)) + 2):
                rt.launch_graph(g)
            reps = max(5, int(min_ms / (per * 20)) + 1)
            rt.record(e0)
            for _ in range(reps):
                rt.launch_graph(g)
            rt.record(e1)
            return rt.elapsed_ms(e0, e1) / (reps * 20) * 1e-3
        except Exception:  # noqa: BLE001
            pass
    for _ in range(int(25.0 / per) + 1):  # >= 25 ms of back-to-back launches: past the chip's clock ramp
        fn()
    iters = max(10, int(min_ms / per) + 1)
    rt.record(e0)
    for _ in range(iters):
        fn()
    rt.record(e1)
    return rt.elapsed_ms(e0, e1) / iters * 1e-3


def cases(ops, rt, big: bool):
    """name -> (callable, algorithmic bytes). Built lazily (one case's tensors at a time)."""
    f16 = torch.float16
    dev = "cuda"

    def rope():
        t = 65536 if big else 2048
        x = torch.randn(1, t, 4096, device=dev).to(f16)
        pos = (torch.arange(t, dtype=torch.int64, device=dev) % 2048)[None]
        y = torch.empty_like(x)
        return (lambda: ops.rope(rt, pos, x, 128, out=y)), 2 * x.numel() * 2, f"{t}x4096 f16 (head dim 128)"

    def rope_headsplit():
        t = 65536 if big else 2048
        x = torch.randn(t // 512, 512, 4096, device=dev).to(f16)
        pos = torch.arange(512, dtype=torch.int64, device=dev).repeat(t // 512, 1)
        y = torch.empty((t // 512, 32, 512, 128), device=dev, dtype=f16)
        return (lambda: ops.rope(rt, pos, x, 128, out=y, head_split=True)), 2 * x.numel() * 2, f"{t}x4096 f16, head-split store"

    def rmsnorm():
        t = 65536 if big else 2048
        x = torch.randn(t, 4096, device=dev).to(f16)
        w = torch.randn(4096, device=dev).to(f16)
        y = torch.empty_like(x)
        return (lambda: ops.rms_norm(rt, x, w, 1e-5, out=y)), 2 * x.numel() * 2, f"{t}x4096 f16"

    def add_rmsnorm():
        t = 65536 if big else 2048
        a = torch.randn(t, 4096, device=dev).to(f16)
        b = torch.randn(t, 4096, device=dev).to(f16)
        w = torch.randn(4096, device=dev).to(f16)
        y = torch.empty_like(a)
        return (lambda: ops.add_layer_norm(rt, a, b, w, None, 1e-5, True, out=y)), 3 * a.numel() * 2, f"{t}x4096 f16 (a + b, then RMSNorm)"

    def gather():
        n = 262144 if big else 16384
        table = torch.randn(30522, 768, device=dev).to(f16)
        idx = torch.randint(0, 30522, (n // 512, 512), device=dev, dtype=torch.int64)
        y = torch.empty((n // 512, 512, 768), device=dev, dtype=f16)
        return (lambda: ops.gather(rt, table, idx, 0, out=y)), 2 * n * 768 * 2 + n * 8, f"{n} rows of 768 f16 from a 30522-row table (int64 indices)"

    def transpose():
        b = 512 if big else 32
        x = torch.randn(b, 512, 12, 64, device=dev).to(f16)
        y = torch.empty((b, 12, 512, 64), device=dev, dtype=f16)
        return (lambda: ops.transpose(rt, x, (0, 2, 1, 3), out=y)), 2 * x.numel() * 2, f"[{b},512,12,64] -> (0,2,1,3) f16"

    def transpose_last():
        b = 512 if big else 32
        x = torch.randn(b, 12, 512, 64, device=dev).to(f16)
        y = torch.empty((b, 12, 64, 512), device=dev, dtype=f16)
        return (lambda: ops.transpose(rt, x, (0, 1, 3, 2), out=y)), 2 * x.numel() * 2, f"[{b},12,512,64] -> (0,1,3,2) f16 (K^T)"

    def add_bias_nchw():
        n = 256 if big else 128
        x = torch.randn(n, 256, 56, 56, device=dev).to(f16)
        b = torch.randn(1, 256, 1, 1, device=dev).to(f16)
        y = torch.empty_like(x)
        return (lambda: ops.binary(rt, "add", x, b, out=y)), 2 * x.numel() * 2, f"[{n},256,56,56] + [1,256,1,1] f16"

    def add():
        n = 256 if big else 128
        x = torch.randn(n, 256, 56, 56, device=dev).to(f16)
        b = torch.randn(n, 256, 56, 56, device=dev).to(f16)
        y = torch.empty_like(x)
        return (lambda: ops.binary(rt, "add", x, b, out=y)), 3 * x.numel() * 2, f"[{n},256,56,56] + same f16"

    def relu():
        n = 256 if big else 128
        x = torch.randn(n, 256, 56, 56, device=dev).to(f16)
        y = torch.empty_like(x)
        return (lambda: ops.unary(rt, "relu", x, out=y)), 2 * x.numel() * 2, f"[{n},256,56,56] f16"

    def gelu():
        n = 131072 if big else 16384
        x = torch.randn(n, 3072, device=dev).to(f16)
        y = torch.empty_like(x)
        return (lambda: ops.unary(rt, "gelu", x, out=y)), 2 * x.numel() * 2, f"[{n},3072] f16"

    def maxpool():
        n = 384 if big else 128
        x = torch.randn(n, 64, 112, 112, device=dev).to(f16)
        y = torch.empty((n, 64, 56, 56), device=dev, dtype=f16)
        return (lambda: ops.max_pool(rt, x, 3, 3, 1, 1, 1, 1, 2, 2, 0, out=y)), (x.numel() + y.numel()) * 2, f"[{n},64,112,112] 3x3/2 pad 1 f16"

    def reduce_mean():
        n = 2560 if big else 128
        x = torch.randn(n, 2048, 7, 7, device=dev).to(f16)
        y = torch.empty((n, 2048, 1, 1), device=dev, dtype=f16)
        return (lambda: ops.reduce(rt, "mean", x, (2, 3), True, out=y)), (x.numel() + y.numel()) * 2, f"[{n},2048,7,7] over (2,3) f16"

    def where():
        n = 65536 if big else 16384
        c = 2048 if big else 768
        x = torch.randn(n, c, device=dev).to(f16)
        y_ = torch.randn(n, c, device=dev).to(f16)
        cond = torch.rand(n, c, device=dev) > 0.5
        o = torch.empty_like(x)
        return (lambda: ops.where(rt, x, y_, cond, out=o)), x.numel() * (2 + 2 + 1 + 2), f"[{n},{c}] f16, bool condition"

    def concat():
        n = 65536 if big else 2048
        a = torch.randn(n, 2048, device=dev).to(f16)
        b = torch.randn(n, 2048, device=dev).to(f16)
        o = torch.empty((n, 4096), device=dev, dtype=f16)
        return (lambda: ops.concat(rt, [a, b], 1, out=o)), 2 * o.numel() * 2, f"2 x [{n},2048] -> axis 1 f16"

    def split():
        n = 65536 if big else 2048
        x = torch.randn(n, 4096, device=dev).to(f16)
        return (lambda: ops.split(rt, x, 1, [2048, 2048])), 2 * x.numel() * 2, f"[{n},4096] -> 2 x [{n},2048] f16"

    def cast():
        n = 65536 if big else 16384
        c = 4096 if big else 768
        x = torch.randn(n, c, device=dev).to(f16)
        y = torch.empty((n, c), device=dev, dtype=torch.float32)
        return (lambda: ops.cast(rt, x, torch.float32, out=y)), x.numel() * 6, f"[{n},{c}] f16 -> f32"

    def silu_mul():
        n = 32768 if big else 2048
        a = torch.randn(n, 11008, device=dev).to(f16)
        b = torch.randn(n, 11008, device=dev).to(f16)
        o = torch.empty_like(a)
        return (lambda: ops.silu_mul(rt, a, b, out=o)), 3 * a.numel() * 2, f"[{n},11008] f16"

    def add_layernorm():
        n = 262144 if big else 16384
        a = torch.randn(n, 768, device=dev).to(f16)
        b = torch.randn(n, 768, device=dev).to(f16)
        g = torch.randn(768, device=dev).to(f16)
        be = torch.randn(768, device=dev).to(f16)
        y = torch.empty_like(a)
        return (lambda: ops.add_layer_norm(rt, a, b, g, be, 1e-12, False, out=y)), 3 * a.numel() * 2, f"[{n},768] f16 (a + b, then LayerNorm)"

    def softmax_of(dt, name):
        def make():
            rows = 1048576 if big else 196608
            x = torch.randn(rows, 512, device=dev).to(dt)
            y = torch.empty_like(x)
            return (lambda: ops.softmax(rt, x, 1, out=y)), 2 * x.numel() * x.element_size(), f"{rows}x512 {name}"
        return make

    def layernorm_of(dt, name):
        def make():
            rows = 524288 if big else 16384
            x = torch.randn(rows, 768, device=dev).to(dt)
            g_, b_ = torch.randn(768, device=dev).to(dt), torch.randn(768, device=dev).to(dt)
            y = torch.empty_like(x)
            return (lambda: ops.layer_norm(rt, x, g_, b_, 1e-5, -1, out=y)), 2 * x.numel() * x.element_size(), f"{rows}x768 {name}"
        return make

    return {"softmax": softmax_of(f16, "f16"), "softmax_f32": softmax_of(torch.float32, "f32"),
            "layernorm": layernorm_of(f16, "f16"), "layernorm_f32": layernorm_of(torch.float32, "f32"),
            "rope": rope, "rope_headsplit": rope_headsplit, "rmsnorm": rmsnorm, "add_rmsnorm": add_rmsnorm, "gather": gather,
            "transpose_0213": transpose, "transpose_0132": transpose_last, "add_bias_nchw": add_bias_nchw, "add": add, "relu": relu,
            "gelu": gelu, "maxpool": maxpool, "reduce_mean": reduce_mean, "where": where, "concat": concat, "split": split, "cast": cast,
            "silu_mul": silu_mul, "add_layernorm": add_layernorm}


def sweep(rt, ops, Event, only=None, budget_s: float | None = None) -> dict:
    import time

    out = {}
    t0 = time.time()
    for big in (False, True):
        for name, make in cases(ops, rt, big).items():
            if only and name not in only:
                continue
            if budget_s is not None and time.time() - t0 > budget_s:
                out.setdefault("skipped", []).append(f"{name}{'_hbm' if big else ''}")
                continue
            try:
                fn, nbytes, shape = make()
                # the inputs are torch kernels on torch's stream; `rt` may launch on a stream of its own (bench.py's does): a Gather
                # that started before its index tensor was written read indices from uninitialised memory — an intermittent
                # HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION that took the whole bench line down (round 4)
                torch.cuda.synchronize()
                t = timeit(rt, Event, fn)
                gbs = nbytes / t / 1e9
                out[f"{name}{'_hbm' if big else ''}"] = {"shape": shape, "us": round(t * 1e6, 2), "MB": round(nbytes / 1e6, 1),
                                                         "GB/s": round(gbs, 1), "frac_hbm_peak": round(gbs / PEAK_HBM_GBS, 4)}
            except Exception as e:  # noqa: BLE001
                out[f"{name}{'_hbm' if big else ''}"] = {"error": repr(e)[:200]}
            torch.cuda.empty_cache()
    return out


def source_stamp() -> str:
    """sha1 over the kernel sources: a counter file is only valid for the kernels it was taken from (bench.py refuses others)."""
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import source_stamps

    return source_stamps.membound_stamp()


def pmc_run(rt, ops, only=None):
    """For tools/profile_membound.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each): every case launched 5 times,
    cases separated by a sentinel launch (a 1-element f32 -> int8 Cast, a kernel no case uses) so that the summariser can cut the
    dispatch list per case without marker tracing. Prints the case list (order, algorithmic bytes) as JSON."""
    listing = []
    s_in = torch.zeros(1, device="cuda")
    s_out = torch.zeros(1, device="cuda", dtype=torch.int8)
    for big in (False, True):
        for name, make in cases(ops, rt, big).items():
            if only and name not in only:
                continue
            fn, nbytes, shape = make()
            torch.cuda.synchronize()
            rt.sync()
            ops.cast(rt, s_in, torch.int8, out=s_out)
            for _ in range(5):
                fn()
            rt.sync()
            listing.append({"case": f"{name}{'_hbm' if big else ''}", "shape": shape, "algorithmic_bytes": nbytes, "launches": 5})
            del fn
            torch.cuda.empty_cache()
    ops.cast(rt, s_in, torch.int8, out=s_out)
    rt.sync()
    print("PMC_CASES " + json.dumps({"stamp": source_stamp(), "cases": listing}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--json", default="")
    ap.add_argument("--once", default="", help="launch this case a few times and exit (for rocprofv3): name or name_hbm")
    ap.add_argument("--pmc-run", action="store_true", help="launch every case 5 times behind a sentinel (tools/profile_membound.sh)")
    args = ap.parse_args()
    from infinitensor_amd import RocmRuntime, ops
    from infinitensor_amd.runtime import Event

    rt = RocmRuntime(0)
    if args.pmc_run or args.once:
        rt.use_torch_stream()
    if args.pmc_run:
        pmc_run(rt, ops, set(args.only.split(",")) if args.only else None)
        return
    if args.once:
        big = args.once.endswith("_hbm")
        fn, nbytes, shape = cases(ops, rt, big)[args.once[:-4] if big else args.once]()
        for _ in range(5):
            fn()
        rt.sync()
        print(json.dumps({"case": args.once, "shape": shape, "algorithmic_bytes": nbytes}))
        return
    res = sweep(rt, ops, Event, set(args.only.split(",")) if args.only else None)
    for k, v in res.items():
        print(f"{k:24s} {json.dumps(v)}", flush=True)
    if args.json:
        Path(args.json).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
