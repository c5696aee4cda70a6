"""MFMA-only ceiling of this MI355X under its power budget (csrc/probe.hip): the headline GEMM's MFMA stream with no
memory or LDS instruction in the loop, on N(0,1) operands (the realistic power draw) and on zeros (the chip clocks
higher). Prints one JSON line.  python tools/mfma_ceiling.py [--dtype bf16] [--iters 4000] [--reps 20]"""
import argparse
import ctypes as C
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from infinitensor_amd import RocmRuntime, lib
from infinitensor_amd._lib import check
from infinitensor_amd.runtime import Event


def mfma_ceiling(rt, dtype=torch.bfloat16, iters=4000, reps=20, fill="normal", shape32=False) -> float:
    """TFLOP/s of the MFMA-only kernel; average over `reps` back-to-back launches (HIP events on the runtime stream)."""
    n = 16 * 512 * 12 * 8
    data = (torch.randn(n, device="cuda") if fill == "normal" else torch.zeros(n, device="cuda")).to(dtype)
    sink = torch.empty(rt.device_info()["compute_units"] * 512, device="cuda", dtype=torch.float32)
    torch.cuda.synchronize()
    code = 16 if dtype == torch.bfloat16 else 10
    flop = C.c_double()

    def launch():
        fn = lib().infini_rocm_probe_mfma_ceiling32 if shape32 else lib().infini_rocm_probe_mfma_ceiling
        check(fn(rt.handle, code, C.c_void_p(data.data_ptr()), C.c_void_p(sink.data_ptr()), iters, C.byref(flop)))

    for _ in range(3):
        launch()
    e0, e1 = Event(), Event()
    rt.record(e0)
    for _ in range(reps):
        launch()
    rt.record(e1)
    s = rt.elapsed_ms(e0, e1) * 1e-3 / reps
    CLOCKS[f"ceiling_{fill}{'_32' if shape32 else ''}"] = core_clock(sink)
    return flop.value / s / 1e12


CLOCKS = {}


def core_clock(sink):
    """(core-clock ticks, 100 MHz ticks) the last launch's workgroup 0 stamped -> MHz of the shader clock inside the loop."""
    torch.cuda.synchronize()
    tc, tr = sink.view(torch.int64)[:2].tolist()
    return round(tc / max(tr, 1) * 100.0, 1)


def a_from_l2(rt, dtype=torch.bfloat16, iters=4000, reps=20, k=512) -> float:
    """TFLOP/s of the MFMA stream with the A fragments loaded straight from an L2-resident panel (no LDS, no B traffic): the upper bound of
    a GEMM that streams one operand L2 -> VGPR (round 6)."""
    a = torch.randn(2048, k, device="cuda").to(dtype)
    bdata = torch.randn(16 * 512 * 8 * 8, device="cuda").to(dtype)
    sink = torch.empty(rt.device_info()["compute_units"] * 512, device="cuda", dtype=torch.float32)
    torch.cuda.synchronize()
    code = 16 if dtype == torch.bfloat16 else 10
    flop = C.c_double()

    def launch():
        check(lib().infini_rocm_probe_mfma_a_from_l2(rt.handle, code, C.c_void_p(a.data_ptr()), C.c_void_p(bdata.data_ptr()), C.c_void_p(sink.data_ptr()),
                                                     k, iters, C.byref(flop)))

    for _ in range(3):
        launch()
    e0, e1 = Event(), Event()
    rt.record(e0)
    for _ in range(reps):
        launch()
    rt.record(e1)
    s = rt.elapsed_ms(e0, e1) * 1e-3 / reps
    return flop.value / s / 1e12


def wave128(rt, dtype=torch.bfloat16, iters=2000, reps=20, pieces=1, zero_tail=False) -> float:
    """TFLOP/s of the 4-wave, 128 x 128-wave-tile form (one wave per SIMD): 128 MFMAs + 32 LDS reads (+ 16 LDS-DMA pieces) per K-tile and wave."""
    panel = torch.randn((16 << 20) // 2 + (512 << 10) // 2, device="cuda").to(dtype)
    if zero_tail:
        panel[(256 << 10) // 2:] = 0
    sink = torch.empty(rt.device_info()["compute_units"] * 256, device="cuda", dtype=torch.float32)
    torch.cuda.synchronize()
    code = 16 if dtype == torch.bfloat16 else 10
    flop = C.c_double()

    def launch():
        check(lib().infini_rocm_probe_mfma_wave128(rt.handle, code, C.c_void_p(panel.data_ptr()), C.c_void_p(sink.data_ptr()), pieces, iters, C.byref(flop)))

    for _ in range(3):
        launch()
    e0, e1 = Event(), Event()
    rt.record(e0)
    for _ in range(reps):
        launch()
    rt.record(e1)
    s = rt.elapsed_ms(e0, e1) * 1e-3 / reps
    CLOCKS[f"wave128_mode{pieces}{'_zeros' if zero_tail else ''}"] = core_clock(sink)
    return flop.value / s / 1e12


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=4000)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
    rt = RocmRuntime(0)
    out = {"dtype": a.dtype, "iters": a.iters, "mfma_per_wave": a.iters * 64,
           "random_TFLOPs": round(mfma_ceiling(rt, dt, a.iters, a.reps, "normal"), 1),
           "zeros_TFLOPs": round(mfma_ceiling(rt, dt, a.iters, a.reps, "zeros"), 1), "nominal_peak_TFLOPs": 2500.0}
    # interleaved repeats of both instruction shapes (the clocks drift over a run)
    for rep in range(3):
        out[f"random_16x16x32_rep{rep}"] = round(mfma_ceiling(rt, dt, a.iters, a.reps, "normal"), 1)
        out[f"random_32x32x16_rep{rep}"] = round(mfma_ceiling(rt, dt, a.iters, a.reps, "normal", shape32=True), 1)
    out["zeros_32x32x16"] = round(mfma_ceiling(rt, dt, a.iters, a.reps, "zeros", shape32=True), 1)
    # the L2 -> VGPR design's upper bound, interleaved with the MFMA-only stream on the same box
    for rep in range(3):
        out[f"a_from_l2_k512_rep{rep}"] = round(a_from_l2(rt, dt, a.iters, a.reps, 512), 1)
        out[f"mfma_only_rep{rep}"] = round(mfma_ceiling(rt, dt, a.iters, a.reps, "normal"), 1)
    out["a_from_l2_k4096"] = round(a_from_l2(rt, dt, a.iters, a.reps, 4096), 1)
    for rep in range(2):
        out[f"wave128_with_dma_rep{rep}"] = round(wave128(rt, dt, a.iters // 2, a.reps, 1), 1)
        out[f"wave128_no_dma_rep{rep}"] = round(wave128(rt, dt, a.iters // 2, a.reps, 0), 1)
        out[f"wave128_with_dma_deep_rep{rep}"] = round(wave128(rt, dt, a.iters // 2, a.reps, 2), 1)
        out[f"wave128_dma_no_reads_rep{rep}"] = round(wave128(rt, dt, a.iters // 2, a.reps, 3), 1)
        out[f"wave128_classic_staging_rep{rep}"] = round(wave128(rt, dt, a.iters // 2, a.reps, 4), 1)
        out[f"wave128_with_dma_deep_skewed_rep{rep}"] = round(wave128(rt, dt, a.iters // 2, a.reps, 5), 1)
        out[f"wave128_dma_aside_random_rep{rep}"] = round(wave128(rt, dt, a.iters // 2, a.reps, 7), 1)
        out[f"wave128_spread_with_dma_rep{rep}"] = round(wave128(rt, dt, a.iters // 2, a.reps, 8), 1)
        out[f"wave128_spread_no_dma_rep{rep}"] = round(wave128(rt, dt, a.iters // 2, a.reps, 9), 1)
        out[f"wave128_dma_aside_zeros_rep{rep}"] = round(wave128(rt, dt, a.iters // 2, a.reps, 7, True), 1)
    out["core_clock_MHz_last_launch"] = CLOCKS
    print(json.dumps(out))
