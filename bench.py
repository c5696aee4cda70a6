#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X operator backend (BASELINE.json metric).

Workload (config.workload): BASELINE.json configs[1] — ONE bf16 MatMul 4096 x 4096 x 4096
(A, B ~ N(0,1) rounded to bf16, fp32 accumulate, bf16 C, no bias, no transpose), resident in HBM.
A "step" is one launch of that MatMul through the C ABI (infini_rocm_matmul) on the runtime's
stream. value = whole-job TFLOP/s = n_gpus * steps * 2*M*N*K / time, weak scaling (each rank owns an
independent 4096^3 problem: the op has no exchange step when sharded by columns — SURVEY 8e).

Before the W warm-up steps the kernel is launched back to back for >= --prewarm-ms (default 40 ms; reported as prewarm_ms /
prewarm_launches): the chip needs ~15 ms to clock up from idle, and a caller that asks for five warm-up steps would otherwise
time the ramp. `roofline.cold20` is the same kernel on an idle chip.

The JSON line is FLAT and small (< 3 KB): besides the contract's fields it carries `config` (workload + every graph / TP /
all-reduce figure BASELINE.json's metric names as a flat scalar: resnet50_bs128_fp16_graph_ms, bert_base_bs32_seq512_fp16_graph_ms,
llama7b_block_tp<N>_ms, allreduce_16MiB_{rccl,direct}_busbw_GBs, tp_overlap_{off,on}_ms, gemm4096_{column,k}_shard_*), `roofline`
(the GEMM + the HBM fractions of Softmax / LayerNorm / the worst memory-bound row) and `cpu_baseline`. The nested detail of the
sections below is written to gpurun_out/bench_detail_n<N>.json (`--print-detail` also prints it in the line):
  roofline      dominant kernel (the GEMM) vs the dense bf16 MFMA peak, timed with HIP events on
                the launch stream inside the timed region; `traffic` from the committed PMC passes, only if they were taken
                from the kernel variant this run launched (config.kernel_variant_launched);
  cpu_baseline  the reference's own native-CPU MatMul (oracle/_ref, built from /root/reference)
                timed on a bounded row-slice of the same problem on this box's host cores
                (rank 0, N=1 only), plus a torch/MKL sgemm figure as the intelcpu-family stand-in;
  tp_block      BASELINE config 5 (one Llama-7B block, Megatron TP over the ranks) incl. `overlap` (reference-shaped
                all-reduce vs row chunks overlapped on a second stream) and, at N > 1, the one-hop reduce-scatter A/B;
  graph_resnet50  BASELINE configs 3 / 4 / 5 through the reference executor + plugin, every graph built in the form and
                operator order pyinfinitensor/onnx.py emits (tools/model_bench.py);
  extras        stand-alone Softmax / LayerNorm HBM-roofline figures (SURVEY 8d C4 shapes), the fp32 MatMul rows against
                the fp32 MFMA peak, the headline GEMM in its other layouts / dtypes.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

M = N = K = 4096
PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16 peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0      # HBM3E spec peak, same guide
PEAK_F32_TFLOPS = 157.3    # fp32 MFMA (v_mfma_f32_32x32x2_f32): 256 CU x 256 FLOP/clk x 2.4 GHz, same guide


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    # default warm-up: 300 launches = 30 ms. The chip needs ~150 back-to-back launches (15 ms) to clock up from idle — measured
    # per launch: 124, 112, 107, 105, 102, 101, 100.6 ... 99.6 us per 25-launch window, flat at 99.5-100 us from launch ~200 to
    # 600 (profiles/r02_launch_series.txt) — and a 20-launch warm-up put that ramp inside the timed region (109.7 us mean).
    ap.add_argument("--warmup", type=int, default=300)
    # Independent of --warmup: back-to-back launches for at least this long before the W warm-up steps, so that a caller who
    # asks for a handful of warm-up steps (the driver: 5) still times the chip at the clocks it holds, not inside its
    # 15 ms ramp from idle. The line reports it (`prewarm_ms`, `prewarm_launches`) next to a COLD figure taken before it
    # (`roofline.cold20`: the first 20 launches on an idle chip).
    ap.add_argument("--prewarm-ms", type=float, default=40.0)
    ap.add_argument("--variant", type=int, default=-1, help="GEMM kernel variant (-1 = heuristic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="skip the ResNet-50 graph-latency row")
    ap.add_argument("--no-tp", action="store_true", help="skip the tensor-parallel block timing")
    ap.add_argument("--print-detail", action="store_true",
                    help="also print the nested sections (tp_block, graph_resnet50, extras) in the line; default: flat scalars in the "
                         "line, the nested detail in gpurun_out/bench_detail_n<N>.json")
    ap.add_argument("--secondary-timeout", type=float, default=420.0,
                    help="seconds the secondary sections (TP block, graphs, CPU baseline, extras) may take before the "
                         "headline line is printed without them and the process ends")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def cpu_baseline_reference() -> dict | None:
    """Runs _cpu_baseline_worker in a fresh interpreter: only one copy of the reference's pybind module can live in a
    process, and this one may already hold the plugin build (graph_resnet50)."""
    import subprocess

    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--cpu-baseline-worker"], capture_output=True, text=True,
                       timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") or ln == "null"]
    if r.returncode != 0 or not lines:
        raise RuntimeError(f"cpu baseline worker failed rc={r.returncode}: {r.stderr[-300:]}")
    return json.loads(lines[-1])


def _cpu_baseline_worker(budget_s: float = 12.0) -> dict | None:
    """Reference native-CPU MatMul (src/kernels/cpu/matmul.cc:6-25, naive ijk, one thread, fp32 — it has
    no bf16 kernel) on the first `rows` rows of the 4096^3 problem, sized to ~budget_s seconds."""
    import importlib.util
    import sysconfig

    import numpy as np

    p = REPO / "oracle" / "_ref" / f"backend{sysconfig.get_config_var('EXT_SUFFIX')}"
    if not p.exists():
        return None
    spec = importlib.util.spec_from_file_location("backend", p)
    backend = importlib.util.module_from_spec(spec)
    sys.modules["backend"] = backend
    spec.loader.exec_module(backend)
    rng = np.random.default_rng(0)
    b = rng.standard_normal((K, N)).astype(np.float32)

    def run(rows: int) -> float:
        a = rng.standard_normal((rows, K)).astype(np.float32)
        h = backend.GraphHandler(backend.cpu_runtime())
        ta, tb = h.tensor([rows, K], 1), h.tensor([K, N], 1)
        h.matmul(ta, tb, None, False, False, None, backend.ActType.Linear, "default")
        h.data_malloc()
        ta.copyin_numpy(a)
        tb.copyin_numpy(b)
        t0 = time.perf_counter()
        h.run()
        return time.perf_counter() - t0

    t_probe = run(8)
    rows = int(max(8, min(1024, 8 * budget_s / max(t_probe, 1e-6))))
    rows -= rows % 8
    t = run(rows)
    flops = 2.0 * rows * N * K
    return {
        "value": flops / t / 1e12,
        "unit": "TFLOP/s",
        "cores": 1,
        "kind": "reference",
        "sample": f"reference native-CPU NaiveMatmul fp32 on rows 0..{rows} of the 4096x4096x4096 problem "
                  f"({flops / 1e9:.1f} GFLOP in {t:.1f} s); host has {os.cpu_count()} cores",
    }


def cpu_standin_mkl() -> dict:
    import torch

    torch.set_num_threads(os.cpu_count() or 1)
    a = torch.randn(M, K)
    b = torch.randn(K, N)
    (a @ b)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        (a @ b)
    t = (time.perf_counter() - t0) / reps
    return {"value": 2.0 * M * N * K / t / 1e12, "unit": "TFLOP/s", "cores": os.cpu_count(),
            "kind": "torch-MKL sgemm fp32 (intelcpu-family stand-in; intelcpu itself is unbuildable here)"}


def extras(rt, ops, Event) -> dict:
    """Stand-alone HBM-bound rows at the SURVEY 8d C4 shapes; algorithmic bytes = 2*numel*sizeof."""
    import torch

    out = {}

    def timeit(fn, iters=50, warm=5):
        # These kernels last 10-300 us and this section runs after a CPU-only pause (the CPU baseline): a handful of
        # warm-up launches leaves the chip at idle clocks (the round-1 driver run read 11.5 us for a LayerNorm that
        # takes 8.5-9.4 us warm, profiles/r02_layernorm_sweep.txt). Warm up for >= 30 ms of back-to-back launches, then
        # time >= 20 ms of them.
        torch.cuda.synchronize()  # the inputs were made by torch on ITS stream; `rt` launches on a stream of its own
        fn()
        rt.sync()
        e0, e1 = Event(), Event()
        rt.record(e0)
        for _ in range(warm):
            fn()
        rt.record(e1)
        per = max(rt.elapsed_ms(e0, e1) / warm, 1e-3)
        for _ in range(int(30.0 / per) + 1):
            fn()
        iters = max(iters, int(20.0 / per) + 1)
        rt.record(e0)
        for _ in range(iters):
            fn()
        rt.record(e1)
        return rt.elapsed_ms(e0, e1) / iters * 1e-3

    # HBM-side bytes per launch (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE, tools/profile_membound.sh). The counter file carries the
    # hash of the kernel sources it was taken from; a file from OTHER sources is refused (traffic: null) — round 3 still quoted
    # r02 counters after rowops.hip had changed.
    sys.path.insert(0, str(REPO / "tools"))
    import membound_sweep as MS

    pmc, pmc_src = {}, None
    for fname in ("r06_membound_pmc.json", "r05_membound_pmc.json", "r04_membound_pmc.json"):
        try:
            d = json.loads((REPO / "profiles" / fname).read_text())
        except Exception as e:  # noqa: BLE001
            pmc_src = pmc_src or ("no counter file: " + repr(e)[:80])
            continue
        if d.get("stamp") == MS.source_stamp():
            pmc, pmc_src = d["rows"], "profiles/" + fname
            break
        pmc_src = "profiles/%s REFUSED: taken from other kernel sources (stamp %s, these are %s)" % (fname, d.get("stamp"), MS.source_stamp())

    def row(kind, shape, name, t, nbytes):
        gbs = nbytes / t / 1e9
        d = {"GB/s": round(gbs, 1), "frac_hbm_peak": round(gbs / PEAK_HBM_GBS, 4), "us": round(t * 1e6, 2)}
        key = {("softmax", "196608x512", "f16"): "softmax", ("softmax", "196608x512", "f32"): "softmax_f32",
               ("layernorm", "16384x768", "f16"): "layernorm", ("layernorm", "16384x768", "f32"): "layernorm_f32"}.get((kind, shape, name))
        r = pmc.get(key) if key else None
        if r:
            d["traffic"] = r["fetch_bytes_x2"] + r["write_bytes"]
            d["traffic_over_algorithmic"] = r["traffic_over_algorithmic"]
        d["traffic_source"] = pmc_src
        return d

    for name, dt in (("f16", torch.float16), ("f32", torch.float32)):
        x = torch.randn(196608, 512, device="cuda").to(dt)
        y = torch.empty_like(x)
        t = timeit(lambda: ops.softmax(rt, x, 1, out=y))
        out[f"softmax_196608x512_{name}"] = row("softmax", "196608x512", name, t, 2 * x.numel() * x.element_size())
        del x, y
        x = torch.randn(16384, 768, device="cuda").to(dt)
        g = torch.randn(768, device="cuda").to(dt)
        b = torch.randn(768, device="cuda").to(dt)
        y = torch.empty_like(x)
        t = timeit(lambda: ops.layer_norm(rt, x, g, b, 1e-5, -1, out=y))
        out[f"layernorm_16384x768_{name}"] = row("layernorm", "16384x768", name, t, 2 * x.numel() * x.element_size())
        # a larger LayerNorm (does not fit the 256 MiB Infinity Cache): 131072 x 768
        x = torch.randn(131072 * 2, 768, device="cuda").to(dt)
        y = torch.empty_like(x)
        t = timeit(lambda: ops.layer_norm(rt, x, g, b, 1e-5, -1, out=y), iters=20)
        out[f"layernorm_262144x768_{name}"] = row("layernorm", "262144x768", name, t, 2 * x.numel() * x.element_size())
        del x, y
    # fp32 MatMul (BASELINE config 1's dtype; the 1e-4 parity gate is stated for it) against the fp32 MFMA peak
    for n, tb in ((512, False), (4096, False), (4096, True)):
        a = torch.randn(n, n, device="cuda")
        b = torch.randn(n, n, device="cuda")
        c = torch.empty(n, n, device="cuda")
        t = timeit(lambda: ops.matmul(rt, a, b, trans_b=tb, out=c), iters=20)
        tf = 2.0 * n ** 3 / t / 1e12
        out[f"matmul_{n}_f32_{'NT' if tb else 'NN'}"] = {"TFLOP/s": round(tf, 2), "frac_fp32_mfma_peak": round(tf / PEAK_F32_TFLOPS, 4),
                                                          "us": round(t * 1e6, 2), "kernel": ops.matmul_last_variant(rt)}
    del a, b, c
    # the headline GEMM in the other layouts / dtypes the reference's MatMul takes (transB = 1 is what ONNX Gemm exports)
    n = 4096
    for name, dt, tb in (("bf16_NT", torch.bfloat16, True), ("f16_NN", torch.float16, False), ("f16_NT", torch.float16, True)):
        a = torch.randn(n, n, device="cuda").to(dt)
        b = torch.randn(n, n, device="cuda").to(dt)
        c = torch.empty(n, n, device="cuda", dtype=dt)
        t = timeit(lambda: ops.matmul(rt, a, b, trans_b=tb, out=c), iters=30)
        out[f"matmul_4096_{name}"] = {"TFLOP/s": round(2.0 * n ** 3 / t / 1e12, 1), "frac_mfma_peak": round(2.0 * n ** 3 / t / 1e12 / PEAK_BF16_TFLOPS, 4), "us": round(t * 1e6, 2)}
    del a, b, c
    torch.cuda.empty_cache()
    # round-5 kernels outside the three BASELINE graphs: the decode-attention step, a depthwise layer, an fp32 3 x 3 layer — timed as
    # hipGraph replays (a ctypes call costs as much as the shorter ones)
    try:
        out["round5_kernels"] = round5_kernels(rt, ops, Event)
    except Exception as e:  # noqa: BLE001
        out["round5_kernels"] = {"error": repr(e)[:200]}
    torch.cuda.empty_cache()
    # every other memory-bound operator of the three graphs at its config shape and at an HBM-sized shape (round-3 verdict #4):
    # RoPE, RMSNorm, Gather, Transpose, broadcast Add, Relu, Gelu, MaxPool, ReduceMean, Where, Concat / Split, Cast, Silu x Mul
    try:
        mb = MS.sweep(rt, ops, Event, budget_s=45.0)
        for k, v in mb.items():
            if isinstance(v, dict) and k in pmc:
                v["traffic_over_algorithmic"] = pmc[k]["traffic_over_algorithmic"]
        out["membound"] = {"rows": mb, "traffic_source": pmc_src,
                           "note": "algorithmic bytes / HIP-event time; *_hbm = a shape beyond the 256 MiB Infinity Cache"}
    except Exception as e:  # noqa: BLE001
        out["membound"] = {"error": repr(e)[:200]}
    return out


def graph_resnet50(local_rank: int, world: int, dist_mod) -> dict:
    """BASELINE config 3: ResNet-50 bs128 fp16 per GPU through the REFERENCE graph executor + the Device::ROCM plugin
    (tools/model_bench.py; hipGraph replay). Data-parallel inference has no collective: N replicas, max-over-ranks."""
    import torch

    sys.path.insert(0, str(REPO / "tools"))
    from model_bench import run_model

    r = run_model("resnet50", local_rank, 128, iters=10, tune=True)
    mm = run_model("matmul", local_rank, dtype="f16", iters=50)  # one-operator graph: executor overhead per launch
    other = {}
    if world == 1:  # BASELINE configs 4 and 5 through the same drop-in path (rank-local figures, N = 1 only)
        # every graph is built in the form and operator order pyinfinitensor/onnx.py emits (tools/model_bench.py); the
        # decomposed row additionally lowers LayerNorm / Gelu to the primitive operators of an opset < 17 export
        for key, model, kw in (("bert_base_bs32_seq512_f16", "bert", {}),
                               ("bert_base_bs32_seq512_f16_decomposed_ln_gelu", "bert", {"decomposed": True}),
                               ("llama7b_block_2048tok_f16_tp1", "llama", {})):
            try:
                m = run_model(model, local_rank, iters=10, **kw)
                other[key] = {k: m[k] for k in ("lowering", "hipgraph_ms", "eager_ms", "hipgraph_TFLOPs", "ops", "fused_launches_per_run", "finite")}
            except Exception as e:  # noqa: BLE001
                other[key] = {"error": repr(e)[:200]}
    ms = torch.tensor([r["hipgraph_ms"], r["eager_ms"]], device="cuda", dtype=torch.float64)
    if world > 1:
        dist_mod.all_reduce(ms, op=dist_mod.ReduceOp.MAX)
    g, e = (float(v) for v in ms.tolist())
    return {"workload": f"ResNet-50 bs128 fp16 per GPU x {world} replicas, reference executor + ROCM plugin, graph in the "
                        f"front-end's lowering ({r['lowering']}: conv -> reshape(bias) -> add), launch planning "
                        + ("on" if r["fusion"] else "off"),
            "fused_launches_per_run": r["fused_launches_per_run"],
            "hipgraph_ms": round(g, 3), "eager_ms": round(e, 3), "samples_per_s": round(world * 128 / g * 1e3, 0),
            "conv_gemm_TFLOPs_aggregate": round(world * r["gemm_conv_TFLOP"] / g * 1e3, 1), "ops": r["ops"], "finite": r["finite"],
            # this rank's figures after the reference's h.tune() (MatMul / Conv choose their kernel by measurement)
            "autotuned": {k: r[k] for k in ("tuned_hipgraph_ms", "tuned_eager_ms", "tune_seconds", "tuned_picks") if k in r},
            "matmul_4096_f16_via_executor": {"eager_ms_incl_sync": mm["eager_ms"], "hipgraph_ms_incl_sync": mm["hipgraph_ms"],
                                             "hipgraph_TFLOPs": mm["hipgraph_TFLOPs"]},
            **other}


def tp_block(rt, ops, Event, world: int, rank: int, dist_mod) -> dict:
    """BASELINE config 5: one Llama-7B-style decoder block (H=4096, 32 heads x 128, FFN 11008), tokens
    B*S = 4*512 = 2048, fp16, Megatron tensor-parallel over `world` ranks (infinitensor_amd/tp.py, mirroring
    examples/distributed/parallel_opt.py): qkv / gate / up column-parallel, o_proj / down row-parallel, ONE
    RCCL all-reduce (16 MiB) after each row-parallel GEMM, on the runtime stream; RoPE (csrc/rope.hip) on q and k.
    Strong scaling: the block is fixed, the weights are sharded. After timing, the same block is run once unsharded
    on this GPU and the max-abs difference to the sharded result is reported (exactly 0 at TP=1)."""
    import torch

    from infinitensor_amd import tp

    Bt, S, H, NH, D, F = 4, 512, 4096, 32, 128, 11008
    T = Bt * S
    nh = NH // world
    dt = torch.float16
    g = torch.Generator(device="cuda").manual_seed(7)  # same full weights on every rank, then sharded
    rnd = lambda *shape: (torch.randn(*shape, device="cuda", generator=g) * 0.02).to(dt)
    x = rnd(T, H)
    full = {"q": rnd(H, H), "k": rnd(H, H), "v": rnd(H, H), "o": rnd(H, H), "g": rnd(H, F), "u": rnd(H, F), "d": rnd(F, H)}
    wq, wk, wv = (tp.shard_column(full[n], world, rank)[0].contiguous() for n in "qkv")
    wo = tp.shard_row(full["o"], world, rank).contiguous()
    wg, wu = (tp.shard_column(full[n], world, rank)[0].contiguous() for n in "gu")
    wd = tp.shard_row(full["d"], world, rank).contiguous()
    pos = torch.arange(S, device="cuda", dtype=torch.int32).repeat(Bt, 1).contiguous()
    n1, n2 = torch.ones(H, device="cuda", dtype=dt), torch.ones(H, device="cuda", dtype=dt)
    scale = torch.full((1,), float(D) ** 0.5, device="cuda", dtype=dt)
    torch.cuda.synchronize()
    # communicator: rank 0 creates the RCCL id, torch.distributed (already up) carries it
    direct_only = os.environ.get("INFINI_ROCM_COMM") == "direct"
    if world > 1 and direct_only:
        # the hand-written transport alone (file rendezvous in a directory every rank of this job derives from MASTER_PORT)
        import tempfile

        rdv = tempfile.gettempdir() + "/irocm_bench_tp_%s" % os.environ.get("MASTER_PORT", "0")
        os.makedirs(rdv, exist_ok=True)
        cwd = os.getcwd()
        os.chdir(rdv)
        try:
            rt.init_comm("bench_tp", world, rank)
        finally:
            os.chdir(cwd)
    elif world > 1:
        idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid = rt.comm_unique_id()
            idt[: len(uid)] = torch.tensor(list(uid), dtype=torch.uint8)
        dist_mod.broadcast(idt, 0)
        rt.init_comm_with_id(bytes(idt.cpu().tolist()), world, rank)
    else:
        rt.init_comm_with_id(rt.comm_unique_id(), 1, 0)

    # Row-parallel GEMM + all-reduce. overlap = 0: the reference's shape — one GEMM, one whole-tensor all-reduce behind it
    # (all_reduce.cc:10-33). overlap = C > 1: the GEMM cut into C row chunks (tokens), each chunk's all-reduce issued on the
    # runtime's comm stream as soon as its GEMM is enqueued, so chunk i's exchange over xGMI runs under chunk i+1's GEMM
    # (infini_rocm_all_reduce_async / comm_join); the consumer waits once. Same operands and sums per output element; a
    # chunk's GEMM may pick another tile / split-K form than the whole GEMM, so results agree to fp16 rounding, not bit for bit.
    def row_parallel(act, w, reduce, overlap):
        out = torch.empty(act.shape[0], w.shape[1], device="cuda", dtype=dt)
        if not reduce or not overlap:
            ops.matmul(rt, act, w, out=out)
            if reduce:
                ops.all_reduce(rt, "sum", out, out=out)
            return out
        rows = act.shape[0] // overlap
        for c in range(overlap):
            sl = slice(c * rows, (c + 1) * rows if c + 1 < overlap else act.shape[0])
            ops.matmul(rt, act[sl], w, out=out[sl])
            ops.all_reduce_async(rt, "sum", out[sl], out=out[sl])
        ops.comm_join(rt)
        return out

    def run_block(w_qkv, w_o, w_gu, w_d, heads_local, reduce, overlap=0):
        def heads(t):  # [T, heads_local*D] -> [Bt*heads_local, S, D]
            return ops.transpose(rt, t.view(Bt, S, heads_local, D), (0, 2, 1, 3)).view(Bt * heads_local, S, D)

        def rope_heads(t):  # rotary embedding per head (head dim 128, theta 1e4: rope.cc:25) with the head split as its store
            return ops.rope(rt, pos, t.view(Bt, S, heads_local * D), D, head_split=True).view(Bt * heads_local, S, D)

        h = ops.rms_norm(rt, x, n1, 1e-5)
        # q, k, v projections of the same activations as ONE grouped launch (batch index = projection, zero A stride: the
        # three weight shards are stacked once at set-up): 3 x 128 tiles of 256^2 leave half the chip idle per launch,
        # 384 tiles in one persistent launch do not (measured 215 -> 169 us at TP = 1)
        qkv = ops.matmul(rt, h, w_qkv)
        q, k = rope_heads(qkv[0]), rope_heads(qkv[1])
        v = heads(qkv[2])
        ctx = ops.attention(rt, q, k, v, scale, scale_is_div=True, head_merge=heads_local).view(T, heads_local * D)
        o = row_parallel(ctx, w_o, reduce, overlap)
        x1 = ops.binary(rt, "add", x, o)
        h2 = ops.rms_norm(rt, x1, n2, 1e-5)
        gu = ops.matmul(rt, h2, w_gu)  # gate and up projections grouped the same way (330 -> 286 us at TP = 1)
        a = ops.silu_mul(rt, gu[0], gu[1])  # Silu -> Mul as one pass (bit-identical to the two kernels)
        d = row_parallel(a, w_d, reduce, overlap)
        return ops.binary(rt, "add", x1, d)

    wqkv, wgu = torch.stack([wq, wk, wv]).contiguous(), torch.stack([wg, wu]).contiguous()
    del wq, wk, wv, wg, wu
    torch.cuda.synchronize()  # torch built the stacks on ITS stream; the block runs on the runtime's own (non-blocking) stream

    def block(overlap=0):
        return run_block(wqkv, wo, wgu, wd, nh, True, overlap)

    e0, e1 = Event(), Event()
    iters = 20

    def time_block(overlap):
        for _ in range(12):  # ~13 ms: keeps the chip on the clocks the headline loop left it at
            yy = block(overlap)
        rt.sync()
        rt.record(e0)
        for _ in range(iters):
            yy = block(overlap)
        rt.record(e1)
        return rt.elapsed_ms(e0, e1) / iters, yy

    ms, y = time_block(0)
    # the same block with the two row-parallel GEMMs cut into 4 row chunks whose all-reduces overlap the next chunk's GEMM
    overlap_info = {"chunks": 4, "off_ms": round(ms, 4), "on_ms": None}
    try:
        ms_on, y_on = time_block(4)
        overlap_info["on_ms"] = round(ms_on, 4)
        overlap_info["max_abs_diff_on_vs_off"] = float((y_on.float() - y.float()).abs().max().item())
        del y_on
    except Exception as e:  # noqa: BLE001  (never take the reference-shaped figure down)
        overlap_info["error"] = repr(e)[:200]
    # parity of the sharded block against the unsharded one on the same GPU (reference launcher: cuda_launch.py:70-76)
    tp_diff = 0.0
    if full is not None:
        # (the stacked unsharded weights are torch kernels on torch's stream: finished before the runtime's stream reads them — round
        # 4: without this wait the unsharded reference was intermittently NaN / off by whole units once N ranks shared one GPU)
        f_qkv, f_gu = torch.stack([full["q"], full["k"], full["v"]]), torch.stack([full["g"], full["u"]])
        torch.cuda.synchronize()
        y_full = run_block(f_qkv, full["o"], f_gu, full["d"], NH, False)
        rt.sync()
        tp_diff = float((y.float() - y_full.float()).abs().max().item())
        if os.environ.get("IROCM_BENCH_TP_DEBUG"):  # which side moved? (both blocks once more; identical launches are bit-identical)
            y_full2 = run_block(f_qkv, full["o"], f_gu, full["d"], NH, False)
            rt.sync()
            print("[tp debug] unsharded run 1 vs 2: %g; sharded vs unsharded 2: %g; nan in sharded %s / unsharded %s %s" % (
                float((y_full.float() - y_full2.float()).abs().max().item()), float((y.float() - y_full2.float()).abs().max().item()),
                bool(torch.isnan(y).any().item()), bool(torch.isnan(y_full).any().item()), bool(torch.isnan(y_full2).any().item())),
                file=sys.stderr, flush=True)
            del y_full2
        del y_full, f_qkv, f_gu
        full = None
    # all-reduce alone: 16 MiB fp16 message
    buf = torch.zeros(T, H, device="cuda", dtype=dt)
    torch.cuda.synchronize()  # (torch's stream -> the runtime's stream)
    for _ in range(3):
        ops.all_reduce(rt, "sum", buf, out=buf)
    rt.record(e0)
    for _ in range(iters):
        ops.all_reduce(rt, "sum", buf, out=buf)
    rt.record(e1)
    ar_ms = rt.elapsed_ms(e0, e1) / iters
    nbytes = buf.numel() * 2
    # strong-scaling shards of the headline GEMM: column shard (no exchange) and K shard (one all-reduce of
    # the 32 MiB bf16 result); at world = 1 both are the plain 4096^3 GEMM
    G = 4096
    bt = torch.bfloat16
    a_full = torch.randn(G, G, device="cuda", generator=g).to(bt)
    b_col = torch.randn(G, G // world, device="cuda", generator=g).to(bt)
    c_col = torch.empty(G, G // world, device="cuda", dtype=bt)
    a_k = torch.randn(G, G // world, device="cuda", generator=g).to(bt)
    b_k = torch.randn(G // world, G, device="cuda", generator=g).to(bt)
    c_k = torch.empty(G, G, device="cuda", dtype=bt)
    torch.cuda.synchronize()

    def col():
        ops.matmul(rt, a_full, b_col, out=c_col)

    def ksh():
        ops.matmul(rt, a_k, b_k, out=c_k)
        ops.all_reduce(rt, "sum", c_k, out=c_k)

    shard_ms = []
    for fn in (col, ksh):
        for _ in range(3):
            fn()
        rt.record(e0)
        for _ in range(iters):
            fn()
        rt.record(e1)
        shard_ms.append(rt.elapsed_ms(e0, e1) / iters)
    tmax = torch.tensor([ms, ar_ms, *shard_ms, overlap_info["on_ms"] or 0.0], device="cuda", dtype=torch.float64)
    if world > 1:
        dist_mod.all_reduce(tmax, op=dist_mod.ReduceOp.MAX)
    ms, ar_ms, col_ms, k_ms, on_ms = (float(v) for v in tmax.tolist())
    overlap_info["off_ms"] = round(ms, 4)
    if overlap_info["on_ms"] is not None:
        overlap_info["on_ms"] = round(on_ms, 4)
    # direct (one-hop, all 7 xGMI links) reduce-scatter + all-gather against RCCL's own all-reduce of the same 16 MiB
    rs_info = None
    if world > 1:
        try:
            xs = torch.zeros(world, T // world, H, device="cuda", dtype=dt)
            sh = torch.empty(T // world, H, device="cuda", dtype=dt)
            torch.cuda.synchronize()
            res = {}
            for direct in (False, True):
                for _ in range(3):
                    ops.reduce_scatter(rt, xs, direct, out=sh)
                    ops.all_gather(rt, sh)
                rt.record(e0)
                for _ in range(iters):
                    ops.reduce_scatter(rt, xs, direct, out=sh)
                    ops.all_gather(rt, sh)
                rt.record(e1)
                res["direct_ms" if direct else "rccl_ms"] = rt.elapsed_ms(e0, e1) / iters
            tm = torch.tensor([res["rccl_ms"], res["direct_ms"]], device="cuda", dtype=torch.float64)
            dist_mod.all_reduce(tm, op=dist_mod.ReduceOp.MAX)
            rs_info = {"reduce_scatter_all_gather_16MiB_rccl_ms": round(float(tm[0]), 4),
                       "reduce_scatter_all_gather_16MiB_direct_ms": round(float(tm[1]), 4)}
        except Exception as e:  # noqa: BLE001
            rs_info = {"error": repr(e)[:200]}
    # the hand-written one-hop transport (csrc/comm_direct.hip: push kernels over IPC-mapped peer buffers, per-workgroup flags)
    # against RCCL on the same 16 MiB all-reduce. Its kernels give up after a time limit instead of hanging (comm_check), and
    # any failure here is reported, never fatal: the RCCL figures above are already taken.
    direct_info = None
    if world > 1 and direct_only:
        direct_info = {"note": "INFINI_ROCM_COMM=direct: every collective of this section already ran on the hand-written transport"}
    elif world > 1:
        d_ms, diff, err = 0.0, 0.0, None
        try:
            os.environ.setdefault("INFINI_ROCM_DIRECT_TIMEOUT_S", "10")
            import tempfile

            rdv = tempfile.gettempdir() + "/irocm_bench_direct_%s" % os.environ.get("MASTER_PORT", "0")
            os.makedirs(rdv, exist_ok=True)
            cwd = os.getcwd()
            os.chdir(rdv)
            try:
                rt.init_comm_direct("bench_direct", world, rank)
            finally:
                os.chdir(cwd)
            rt.comm_set_algo(1)
            for _ in range(3):
                ops.all_reduce(rt, "sum", buf, out=buf)
            rt.sync()
            rt.comm_check()  # a transport that does not work on this node is reported after ONE time limit, not timed
            rt.record(e0)
            for _ in range(iters):
                ops.all_reduce(rt, "sum", buf, out=buf)
            rt.record(e1)
            d_ms = rt.elapsed_ms(e0, e1) / iters
            # parity of the two transports on fresh data: same sum (fp32 accumulation in rank order vs RCCL's order: f16 rounding)
            probe = (torch.randn(1 << 20, device="cuda", generator=g) * 0.1).to(dt)
            torch.cuda.synchronize()
            got_d = ops.all_reduce(rt, "sum", probe)
            rt.comm_set_algo(0)
            got_r = ops.all_reduce(rt, "sum", probe)
            rt.sync()
            rt.comm_check()
            diff = float((got_d.float() - got_r.float()).abs().max().item())
        except Exception as e:  # noqa: BLE001
            err = repr(e)[:300]
            try:
                rt.comm_set_algo(0)
            except Exception:  # noqa: BLE001
                pass
        # every rank reaches this collective whatever happened above (a rank that failed must not leave the others waiting)
        tm = torch.tensor([0.0 if err else 1.0, -d_ms, -diff], device="cuda", dtype=torch.float64)
        dist_mod.all_reduce(tm, op=dist_mod.ReduceOp.MIN)
        if float(tm[0]) > 0.5:
            d_ms = -float(tm[1])
            direct_info = {"allreduce_direct_ms": round(d_ms, 4),
                           "allreduce_direct_busbw_GBs": round(2 * (world - 1) / world * nbytes / (d_ms * 1e-3) / 1e9, 1),
                           "max_abs_diff_vs_rccl": -float(tm[2]),
                           "note": "reduce-scatter + all-gather as ONE push kernel per rank (32 workgroups), fp32 sums in rank order"}
        else:
            direct_info = {"error": err or "another rank failed"}
    gemm_shards = {
        "workload": "one bf16 4096^3 GEMM strong-scaled over %d GPUs" % world,
        "column_shard_ms": round(col_ms, 4), "column_shard_TFLOPs_aggregate": round(2.0 * G ** 3 / col_ms / 1e9, 1),
        # a K shard needs world > 1 (at world 1 it is the plain GEMM and the all-reduce is an identity copy)
        "k_shard_allreduce_ms": round(k_ms, 4) if world > 1 else None,
        "k_shard_TFLOPs_aggregate": round(2.0 * G ** 3 / k_ms / 1e9, 1) if world > 1 else None,
    }
    return {
        "workload": "Llama-7B-style block (RMSNorm, q/k/v, RoPE, attention, o, RMSNorm, gate/up/SiLU, down), tokens 2048, fp16, TP=%d (2 all-reduces of 16 MiB)" % world,
        "ms_per_block": round(ms, 4),
        "gemm_TFLOPs_aggregate": round(tp.llama_block_flops(T, H, F, 1) / ms / 1e9, 1),
        "allreduce_16MiB_ms": round(ar_ms, 4) if world > 1 else None,  # an identity copy at world 1: not a number
        "allreduce_busbw_GBs": round(2 * (world - 1) / world * nbytes / (ar_ms * 1e-3) / 1e9, 1) if world > 1 else None,
        # row-parallel GEMMs in 4 row chunks, each chunk's all-reduce on the comm stream under the next chunk's GEMM
        "overlap": overlap_info,
        "reduce_scatter_all_gather": rs_info,
        "allreduce_direct": direct_info,
        "gemm_strong_scaling": gemm_shards,
        "max_abs_diff_vs_unsharded": tp_diff,
        "finite": bool(torch.isfinite(y.float()).all().item()),
    }


def round5_kernels(rt, ops, Event) -> dict:
    """AttentionKVCache decode step (B x H = 32, 4096 cached keys, D = 128, f16: a batch-1 Llama-7B step; algorithmic bytes = K and V
    once), one depthwise layer of EfficientNet-Lite4 (C144 150 x 150 3 x 3 / 2 at batch 32 f16; input + output once) and one fp32
    3 x 3 layer of ResNet-50 at batch 32 (C128 28 x 28; against the 157 TF/s fp32 MFMA peak)."""
    import torch

    def graph_us(launch, iters):
        for i in range(3):
            launch(i)
        rt.sync()
        rt.begin_capture()
        for i in range(iters):
            launch(i)
        g = rt.end_capture()
        rt.launch_graph(g)
        e0, e1 = Event(), Event()
        rt.record(e0)
        rt.launch_graph(g)
        rt.record(e1)
        rt.sync()
        return rt.elapsed_ms(e0, e1) / iters * 1e3

    out = {}
    bh, n, d = 32, 4096, 128
    sets = 8  # rotated: one set (67 MB) would sit in the 256 MiB Infinity Cache
    kc = [torch.randn(1, bh, n, d, device="cuda").half() for _ in range(sets)]
    vc = [torch.randn(1, bh, n, d, device="cuda").half() for _ in range(sets)]
    q, k, v = (torch.randn(1, bh, 1, d, device="cuda").half() for _ in range(3))
    pos = torch.tensor([n - 1], dtype=torch.int32, device="cuda")
    o = torch.empty_like(q)
    torch.cuda.synchronize()
    us = graph_us(lambda i: ops.attention_kvcache(rt, kc[i % sets], vc[i % sets], q, k, v, pos, out=o), 40)
    nbytes = 2.0 * bh * n * d * 2
    out["decode_attention_bh32_n4096_d128_f16"] = {"us": round(us, 2), "frac_hbm_peak": round(nbytes / us / 1e3 / PEAK_HBM_GBS, 4)}
    del kc, vc
    c, h = 144, 150
    x = torch.randn(32, c, h, h, device="cuda").half()
    w = (torch.randn(c, 1, 3, 3, device="cuda") / 3).half()
    b = torch.randn(c, device="cuda").half()
    y = torch.empty(32, c, 75, 75, device="cuda", dtype=torch.float16)
    torch.cuda.synchronize()
    us = graph_us(lambda i: ops.conv2d(rt, x, w, 1, 1, 2, 2, bias=b, act=1, out=y), 20)
    nbytes = 2.0 * 32 * c * (h * h + 75 * 75)
    out["depthwise_c144_150x150_3x3_s2_bs32_f16"] = {"us": round(us, 2), "frac_hbm_peak": round(nbytes / us / 1e3 / PEAK_HBM_GBS, 4),
                                                     "route": ops.conv_last_route(rt)}
    del x, w, b, y
    x = torch.randn(32, 128, 28, 28, device="cuda")
    w = torch.randn(128, 128, 3, 3, device="cuda") / 34.0
    b = torch.randn(128, device="cuda")
    y = torch.empty(32, 128, 28, 28, device="cuda")
    torch.cuda.synchronize()
    ops.set_conv_const_weights(rt, True)  # graph weights, as the plugin declares them: the tap-major image is packed once, not per call
    try:
        us = graph_us(lambda i: ops.conv2d(rt, x, w, 1, 1, 1, 1, bias=b, act=1, out=y), 10)
    finally:
        ops.set_conv_const_weights(rt, False)
    tf = 2.0 * 32 * 128 * 28 * 28 * 128 * 9 / us / 1e6
    out["conv_fp32_c128_28x28_3x3_bs32"] = {"us": round(us, 2), "TFLOP/s": round(tf, 1), "frac_fp32_mfma_peak": round(tf / PEAK_F32_TFLOPS, 4),
                                            "route": ops.conv_last_route(rt)}
    return out


def flat_summary(line: dict, detail: dict, world: int) -> None:
    """Copy every figure BASELINE.json's metric / north_star names out of the nested sections into FLAT scalars of `config` /
    `roofline` (the objects the driver's record keeps). tests/test_bench_line_cpu.py runs it on round 4's nested line."""
    cfg, roof = line["config"], line["roofline"]
    g = detail.get("graph_resnet50") or {}
    if "hipgraph_ms" in g:
        cfg["resnet50_bs128_fp16_graph_ms"] = g["hipgraph_ms"]
        cfg["resnet50_bs128_fp16_eager_ms"] = g.get("eager_ms")
        cfg["resnet50_bs128_fp16_samples_per_s"] = g.get("samples_per_s")
        cfg["resnet50_launches_per_run"] = g.get("fused_launches_per_run")
        t = (g.get("autotuned") or {}).get("tuned_hipgraph_ms")
        if t is not None:
            cfg["resnet50_bs128_fp16_graph_ms_after_tune"] = t
    elif "error" in g:
        cfg["resnet50_error"] = str(g["error"])[:100]
    for key, short in (("bert_base_bs32_seq512_f16", "bert_base_bs32_seq512_fp16_graph_ms"),
                       ("bert_base_bs32_seq512_f16_decomposed_ln_gelu", "bert_base_decomposed_graph_ms"),
                       ("llama7b_block_2048tok_f16_tp1", "llama7b_block_tp1_via_executor_graph_ms")):
        m = g.get(key) or {}
        if "hipgraph_ms" in m:
            cfg[short] = m["hipgraph_ms"]
    tpb = detail.get("tp_block") or {}
    if "ms_per_block" in tpb:
        cfg[f"llama7b_block_tp{world}_ms"] = tpb["ms_per_block"]
        cfg[f"llama7b_block_tp{world}_TFLOPs_aggregate"] = tpb.get("gemm_TFLOPs_aggregate")
        cfg["tp_max_abs_diff_vs_unsharded"] = tpb.get("max_abs_diff_vs_unsharded")
        ov = tpb.get("overlap") or {}
        cfg["tp_overlap_off_ms"], cfg["tp_overlap_on_ms"] = ov.get("off_ms"), ov.get("on_ms")
        cfg["allreduce_16MiB_rccl_ms"] = tpb.get("allreduce_16MiB_ms")
        cfg["allreduce_16MiB_rccl_busbw_GBs"] = tpb.get("allreduce_busbw_GBs")
        d = tpb.get("allreduce_direct") or {}
        cfg["allreduce_16MiB_direct_ms"] = d.get("allreduce_direct_ms")
        cfg["allreduce_16MiB_direct_busbw_GBs"] = d.get("allreduce_direct_busbw_GBs")
        rs = tpb.get("reduce_scatter_all_gather") or {}
        cfg["rs_ag_16MiB_rccl_ms"] = rs.get("reduce_scatter_all_gather_16MiB_rccl_ms")
        cfg["rs_ag_16MiB_direct_ms"] = rs.get("reduce_scatter_all_gather_16MiB_direct_ms")
        sh = tpb.get("gemm_strong_scaling") or {}
        cfg["gemm4096_column_shard_ms"] = sh.get("column_shard_ms")
        cfg["gemm4096_column_shard_TFLOPs_aggregate"] = sh.get("column_shard_TFLOPs_aggregate")
        cfg["gemm4096_k_shard_allreduce_ms"] = sh.get("k_shard_allreduce_ms")
        cfg["gemm4096_k_shard_TFLOPs_aggregate"] = sh.get("k_shard_TFLOPs_aggregate")
    elif "error" in tpb:
        cfg["tp_block_error"] = str(tpb["error"])[:100]
    ex = detail.get("extras") or {}
    for key, short in (("softmax_196608x512_f16", "softmax_196608x512_f16_frac_hbm"), ("layernorm_16384x768_f16", "layernorm_16384x768_f16_frac_hbm"),
                       ("layernorm_262144x768_f16", "layernorm_262144x768_f16_frac_hbm")):
        if key in ex and "frac_hbm_peak" in ex[key]:
            roof[short] = ex[key]["frac_hbm_peak"]
            # HBM-side bytes over algorithmic bytes from the source-stamped counter passes (null when the committed file was taken from
            # other kernel sources: `traffic_source` in the detail file says which)
            if key != "layernorm_262144x768_f16": # (the counter passes cover the two config shapes; the HBM-sized shapes are in extras.membound)
                roof[short.replace("_frac_hbm", "_traffic_over_algorithmic")] = ex[key].get("traffic_over_algorithmic")
    if "matmul_4096_f32_NN" in ex:
        roof["matmul_4096_f32_frac_fp32_mfma_peak"] = ex["matmul_4096_f32_NN"].get("frac_fp32_mfma_peak")
    r5 = ex.get("round5_kernels") or {}
    for key, short, field in (("decode_attention_bh32_n4096_d128_f16", "decode_attn_bh32_n4096_f16_frac_hbm", "frac_hbm_peak"),
                              ("depthwise_c144_150x150_3x3_s2_bs32_f16", "dwconv_c144_150_s2_f16_frac_hbm", "frac_hbm_peak"),
                              ("conv_fp32_c128_28x28_3x3_bs32", "conv_fp32_c128_28_3x3_bs32_frac_fp32_mfma", "frac_fp32_mfma_peak")):
        if key in r5 and field in r5[key]:
            roof[short] = r5[key][field]
    rows = (ex.get("membound") or {}).get("rows") or {}
    hbm = {k: v["frac_hbm_peak"] for k, v in rows.items() if isinstance(v, dict) and k.endswith("_hbm") and "frac_hbm_peak" in v}
    if hbm:
        worst = min(hbm, key=hbm.get)
        roof["membound_hbm_rows"] = len(hbm)
        roof["membound_hbm_min_frac"] = hbm[worst]
        roof["membound_hbm_min_row"] = worst
        roof["membound_hbm_rows_below_0p60"] = sum(1 for v in hbm.values() if v < 0.60)
    for k in ("softmax_hbm", "layernorm_hbm", "reduce_mean_hbm", "maxpool_hbm"):
        if k in hbm:
            roof[k + "_frac"] = hbm[k]


def pmc_traffic(launched: str) -> dict:
    """HBM-side bytes per GEMM launch from the committed PMC passes (tools/profile_gemm.sh: rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE of the same shape; FETCH_SIZE x2 per the gfx950 note). The counter file names the kernel variant it was
    taken from; a file from ANOTHER variant than the one this run launched is refused (traffic: null) instead of silently
    going stale when the kernel changes."""
    sys.path.insert(0, str(REPO / "tools"))
    import source_stamps

    refusal = None
    for name in ("r06_gemm_pmc.json", "r06_gemm256p_pmc.json", "r05_gemm256p_pmc.json", "r04_gemm256p_pmc.json", "r03_gemm256p_pmc.json", "r02_gemm256p_pmc.json"):
        p = REPO / "profiles" / name
        try:
            d = json.loads(p.read_text())
        except Exception:  # noqa: BLE001
            continue
        # (round 6: the file carries the hash of the GEMM sources it measured; files of earlier rounds have none and are judged by the
        # variant name alone — their kernel's K loop is unchanged, its prologue is not)
        if d.get("stamp") and d["stamp"] != source_stamps.gemm_stamp():
            return {"traffic": None, "traffic_source": f"profiles/{name} REFUSED: taken from other GEMM sources (stamp {d['stamp']}, these are {source_stamps.gemm_stamp()})"}
        try:
            from infinitensor_amd import ops

            file_variant = d.get("variant_name") or ops.matmul_variants()[int(d["variant"])]
        except Exception:  # noqa: BLE001
            file_variant = None
        src = f"profiles/{name} (variant {file_variant})"
        if file_variant != launched:
            # (round 6 keeps one file per GEMM kernel: the next name may be the launched one's)
            refusal = refusal or {"traffic": None, "traffic_source": f"{src} REFUSED: this run launched {launched}"}
            continue
        return {"traffic": float(d["traffic_bytes_per_launch"]), "traffic_source": src}
    return refusal or {"traffic": None, "traffic_source": "no counter file under profiles/"}


# Test mode (tests/test_gpu_multi.py): IROCM_BENCH_ONE_DEVICE=1 with INFINI_ROCM_COMM=direct runs the N ranks on ONE device — the
# hand-written transport allows what RCCL refuses — with torch.distributed on gloo. The figures are meaningless as scaling
# numbers (the ranks share a GPU); what it checks is that the tensor-parallel block at world N equals the unsharded block.
ONE_DEVICE = os.environ.get("IROCM_BENCH_ONE_DEVICE") == "1"


def self_spawn(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in the environment — exactly what `python -m torch.distributed.run` would set) and relay
    rank 0's output. Mirrors the reference launcher, which spawns its own workers
    (examples/distributed/cuda/cuda_launch.py:110-130)."""
    import socket
    import subprocess

    import torch

    have = torch.cuda.device_count()
    if have < args.gpus and not ONE_DEVICE:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(REPO / "bench.py"), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0)
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def main() -> int:
    args = parse()
    if args.cpu_baseline_worker:
        print(json.dumps(_cpu_baseline_worker()), flush=True)
        return 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_spawn(args)
    import torch

    from infinitensor_amd import RocmRuntime, ops
    from infinitensor_amd.runtime import Event

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = world > 1
    if dist:
        import torch.distributed as td

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if ONE_DEVICE:
            local_rank = 0
            torch.cuda.set_device(0)
            td.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            td.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    assert torch.cuda.is_available(), "bench.py needs a GPU; the HIP path has no fallback"

    rt = RocmRuntime(local_rank)  # own non-blocking stream; kernels are timed on THIS stream
    info = rt.device_info()
    gen = torch.Generator(device="cuda").manual_seed(1234 + rank)
    a = torch.randn(M, K, device="cuda", generator=gen).to(torch.bfloat16)
    b = torch.randn(K, N, device="cuda", generator=gen).to(torch.bfloat16)
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    torch.cuda.synchronize()
    if args.variant >= 0:
        ops.set_matmul_variant(rt, args.variant)

    def step():
        ops.matmul(rt, a, b, out=c)

    # cold figure: 20 launches on an IDLE chip (code objects loaded by two untimed launches, then 0.25 s of nothing: the
    # clocks are back down)
    step()
    step()
    rt.sync()
    time.sleep(0.25)
    ec0, ec1 = Event(), Event()
    rt.record(ec0)
    for _ in range(20):
        step()
    rt.record(ec1)
    rt.sync()
    cold_us = rt.elapsed_ms(ec0, ec1) * 1e3 / 20
    kernel_variant_launched = ops.matmul_last_variant(rt)
    # time-based pre-warm (not part of --warmup, stated in the line)
    prewarm_launches = 0
    t_pw = time.perf_counter()
    while (time.perf_counter() - t_pw) * 1e3 < args.prewarm_ms:
        for _ in range(16):
            step()
        prewarm_launches += 16
        rt.sync()
    prewarm_ms = (time.perf_counter() - t_pw) * 1e3
    for _ in range(args.warmup):
        step()
    rt.sync()

    def barrier():
        if dist:
            td.barrier()

    e0, e1 = Event(), Event()
    barrier()
    torch.cuda.synchronize()
    rt.sync()
    t0 = time.perf_counter()
    rt.record(e0)
    for i in range(args.steps):
        step()
    rt.record(e1)
    rt.sync()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_s = rt.elapsed_ms(e0, e1) * 1e-3 / args.steps  # avg launch duration from HIP events over the timed region
    # Per-launch spread, from a SECOND pass of the same launches with an event after every one (<= 1000 steps): it shows
    # the clock state over a run (the chip ramps UP over its first ~150 launches from idle and then holds). Kept out of the
    # timed region: every event record is a marker packet between two dispatches and costs 2-6 us of queue time per launch
    # (measured: 101.6 us per launch with the markers against 95.7 us kernel duration in the rocprofv3 trace of the same run;
    # 96.8 against 94.8 us on another box).
    per_launch = None
    if args.steps <= 1000:
        marks = [Event() for _ in range(args.steps)]
        m0 = Event()
        rt.record(m0)
        for i in range(args.steps):
            step()
            rt.record(marks[i])
        rt.sync()
        prev, durs = m0, []
        for m in marks:
            durs.append(rt.elapsed_ms(prev, m) * 1e3)
            prev = m
        durs_sorted = sorted(durs)
        per_launch = {"min": round(durs_sorted[0], 2), "median": round(durs_sorted[len(durs) // 2], 2),
                      "mean": round(sum(durs) / len(durs), 2), "max": round(durs_sorted[-1], 2),
                      "first10_mean": round(sum(durs[:10]) / min(10, len(durs)), 2),
                      "last10_mean": round(sum(durs[-10:]) / min(10, len(durs)), 2),
                      "note": "second pass, one event record after every launch (each adds 2-6 us of queue time)"}

    if dist:
        tmax = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        elapsed = float(tmax.item())

    flop_per_step = 2.0 * M * N * K
    value = world * args.steps * flop_per_step / elapsed / 1e12
    achieved = flop_per_step / kernel_s / 1e12

    # The headline part of the line is complete here. Everything below is secondary (TP block, ResNet-50 graph, CPU
    # baseline, extras): a watchdog prints the line with what has been gathered and ends the process if a secondary
    # section stalls (e.g. a rank lost inside a collective), so the headline measurement always reaches the driver.
    line = {
        "metric": "bf16 MatMul 4096x4096x4096 throughput (GEMM TFLOP/s, % dense MFMA peak)",
        "value": round(value, 2),
        "unit": "TFLOP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "prewarm_ms": round(prewarm_ms, 1),
        "prewarm_launches": prewarm_launches,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic (N(0,1) rounded to bf16, seeded)",
        "config": {
            "workload": "BASELINE configs[1]: one bf16 MatMul M=N=K=4096 per GPU, NN layout, fp32 accumulate, via infini_rocm_matmul",
            "kernel_variant": ops.matmul_variants()[args.variant] if args.variant >= 0 else "heuristic",
            "kernel_variant_launched": kernel_variant_launched,
            "device": info["name"], "arch": info["arch"], "compute_units": info["compute_units"],
            "clock_mhz": info["clock_mhz"],
            "parallelism": f"{world} independent replicas (column-sharded GEMM has no collective)",
        },
        "roofline": {
            "bound": "mfma",
            "achieved": round(achieved, 2),
            "peak": PEAK_BF16_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
            **pmc_traffic(kernel_variant_launched),
            "kernel_us": round(kernel_s * 1e6, 3),
            # the same kernel on an idle chip: 20 launches after 0.25 s of idleness, before any warm-up
            "cold20": {"kernel_us": round(cold_us, 3), "achieved": round(flop_per_step / (cold_us * 1e-6) / 1e12, 2),
                       "frac": round(flop_per_step / (cold_us * 1e-6) / 1e12 / PEAK_BF16_TFLOPS, 4)},
            "cold20_kernel_us": round(cold_us, 3),
            "cold20_frac": round(flop_per_step / (cold_us * 1e-6) / 1e12 / PEAK_BF16_TFLOPS, 4),
            "kernel_us_per_launch": per_launch,
            "peak_from_device": round(info["compute_units"] * 4096 * info["clock_mhz"] * 1e6 / 1e12, 1),
        },
    }
    import ctypes
    import threading

    printed = threading.Lock()
    # The printed line stays SMALL (< 3 KB) and FLAT: the driver's record keeps the top-level scalars and the `config` /
    # `roofline` / `cpu_baseline` objects (scalars only) and the last ~2 KB of stdout. Round 4 printed the ResNet-50 / TP-block
    # results as nested objects in front of 9 KB of memory-bound rows, and the second half of BASELINE's metric (ResNet-50 bs128
    # graph ms) never reached BENCH_r04.json. Now: every figure BASELINE.json's metric / north_star names is a flat scalar inside
    # `config` (graph latencies, TP block, all-reduce bus bandwidth) or `roofline` (HBM fractions of Softmax / LayerNorm); the
    # nested detail of every section goes to a FILE (gpurun_out/bench_detail_n<N>.json, path in the line).
    detail: dict = {}

    def emit(extra: dict | None = None) -> None:
        if not printed.acquire(blocking=False):
            return
        if rank == 0:
            if extra:
                line.update(extra)
            try:
                flat_summary(line, detail, world)
            except Exception as e:  # noqa: BLE001  (the headline line must go out whatever a secondary section returned)
                line["summary_error"] = repr(e)[:120]
            try:
                out_dir = REPO / "gpurun_out"
                out_dir.mkdir(exist_ok=True)
                df = out_dir / f"bench_detail_n{world}.json"
                df.write_text(json.dumps({"line": line, **detail}, indent=1))
                line["detail_file"] = str(df.relative_to(REPO))
            except Exception as e:  # noqa: BLE001
                line["detail_file"] = "not written: " + repr(e)[:80]
            if args.print_detail:
                line.update(detail)
            else:
                # nested members and long provenance strings of the kept objects live in the detail file only; `config` goes LAST
                # so that the graph / TP scalars are also inside the last 2 KB of stdout
                for obj in ("roofline", "cpu_baseline"):
                    if isinstance(line.get(obj), dict):
                        line[obj] = {k: v for k, v in line[obj].items() if not isinstance(v, (dict, list)) and k != "attainable_peak_source"}
                line["config"] = line.pop("config")
            ctypes.CDLL(None).fflush(None)  # RCCL prints a version banner through C stdio: keep the JSON line last
            print(json.dumps(line), flush=True)

    def watchdog():
        emit({"secondary_timeout_s": args.secondary_timeout})
        os._exit(0)

    wd = threading.Timer(args.secondary_timeout, watchdog)
    wd.daemon = True
    wd.start()

    try:  # the measured denominator: what the matrix pipes sustain on this chip with no memory traffic at all
        sys.path.insert(0, str(REPO / "tools"))
        from mfma_ceiling import mfma_ceiling

        att = mfma_ceiling(rt, torch.bfloat16, iters=2000, reps=10, fill="normal")
        line["roofline"]["attainable_peak"] = round(att, 1)
        line["roofline"]["frac_of_attainable"] = round(achieved / att, 4)
        line["roofline"]["attainable_peak_source"] = ("MFMA-only kernel (csrc/probe.hip: the GEMM's v_mfma_f32_16x16x32 stream, 8 waves "
                                                      "per CU, N(0,1) bf16 operands, no memory / LDS instructions), same process")
    except Exception as e:  # noqa: BLE001
        line["roofline"]["attainable_peak"] = None
        line["roofline"]["attainable_peak_error"] = repr(e)[:200]

    def note(what):  # progress on stderr: which secondary section a fault or a stall belongs to
        print(f"[bench] rank {rank}: {what}", file=sys.stderr, flush=True)

    if not args.no_tp:
        note("tp_block")
        try:  # never let the secondary measurement take the headline line down
            detail["tp_block"] = tp_block(rt, ops, Event, world, rank, td if dist else None)
        except Exception as e:  # noqa: BLE001
            detail["tp_block"] = {"error": repr(e)[:300]}

    if not args.no_graph:
        note("graph_resnet50")
        try:
            detail["graph_resnet50"] = graph_resnet50(local_rank, world, td if dist else None)
        except BaseException as e:  # noqa: BLE001  (SystemExit when the plugin build is absent)
            detail["graph_resnet50"] = {"error": repr(e)[:300]}

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            note("cpu_baseline")
            try:
                cb = cpu_baseline_reference()
            except Exception as e:  # keep the bench line even if the oracle module is missing
                cb = {"error": repr(e)}
            try:
                mkl = cpu_standin_mkl()
            except Exception as e:
                mkl = {"error": repr(e)}
            if cb is not None:
                # top-level value / cores / kind / sample = the reference's own CPU kernel (the contract's fields);
                # both CPU figures of BASELINE.md section 3 sit inside this one object, cores stated for each
                cb = dict(cb)
                cb["reference_1core"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample") if k in cb}
                cb[f"mkl_standin_{mkl.get('cores', os.cpu_count())}core"] = mkl
                cb["host_cores"] = os.cpu_count()
                if "value" in mkl:  # (flat copies: the driver's record keeps scalars only)
                    cb["mkl_standin_TFLOPs"] = round(mkl["value"], 3)
                    cb["mkl_standin_cores"] = mkl.get("cores")
                line["cpu_baseline"] = cb
            else:
                line["cpu_baseline"] = {"error": "oracle/_ref is not built", f"mkl_standin_{os.cpu_count()}core": mkl}
        if world == 1 and not args.no_extras:
            note("extras")
            try:
                detail["extras"] = extras(rt, ops, Event)
            except Exception as e:
                detail["extras"] = {"error": repr(e)}
    note("done")
    emit()
    if dist:  # the watchdog stays armed: a rank that never reaches this barrier must not hold the others
        td.barrier()
        td.destroy_process_group()
    wd.cancel()
    return 0


if __name__ == "__main__":
    sys.exit(main())
