// Depthwise Conv2d (groups == C, one input channel per filter; filters per channel = F / C >= 1) for f16 / bf16, NCHW x F1RS.
//
// Reference semantics: src/operators/conv.cc:47-114 (groups = C / channel_per_group; a depthwise layer is the case
// channel_per_group == 1), kernel src/kernels/cuda/conv.cc:57-168 (cuDNN group convolution), index math as the native CPU kernel
// src/kernels/cpu/conv.cc:25-50. EfficientNet-Lite4 — one of the CNNs the reference lists as validated
// (docs/SUPPORT_MATRIX_CN.md:24-27) — is built from 3 x 3 and 5 x 5 depthwise layers of stride 1 and 2. Until round 5 they fell
// through every fast route (all of which require groups == 1) to the generic implicit GEMM per (image, group) with K = R S = 9 / 25:
// a 128 x 128 x 32 MFMA tile for ONE filter row (C32 150 x 150 at batch 32: 2.2 ms for 92 MB).
//
// A depthwise layer has no contraction to speak of (2 R S FLOP per output element, 18-50): it is bound by moving the input and the
// output once through HBM, and after that by the vector ALU (R S fp32 FMAs per output). Kernel: one thread per (output plane, column
// strip of 8 outputs, row strip of TH output rows). The thread walks DOWN its strip: per output row it fetches the sh new input rows
// of its window — 8 sw + S - 1 columns, as 16-byte loads at 2-byte alignment, neighbours' loads hit L1 / L2 — masks the columns
// outside the image row (AND on the packed data; the masks are set up once per thread), converts to fp32 into a ring of R register
// rows, and accumulates the 8 outputs in the oracle's tap order (r outer, s inner). The ring is indexed statically (the row loop is
// unrolled R times: R sh = 0 mod R), so it lives in registers. Rows above / below the image arrive as zeros (the load is pointed past
// the buffer descriptor's range). No LDS, no barrier, everything a thread needs per strip (item decode, R S weights, bias, masks)
// is set up once. (Version 1 of this file staged rows through LDS and decoded one item per 8 outputs: 4.5 VALU-issue cycles per
// instruction with 78 % of the instructions overhead — 0.12-0.35 of the HBM peak.)
#include "gemm256_common.h" // (sfor: compile-time loops)
#include <algorithm>

namespace irocm {
using g256::sfor;

struct DwArgs {
    const void *x, *w, *bias;
    void *y;
    int planes_out;   // N * F
    int c, f, mult;   // mult = F / C filters per input channel: filter fi reads channel fi / mult
    int h, wd, oh, ow;
    int ph, pw, sh, sw;
    int th;           // output rows per thread
    int strips;       // ceil(oh / th)
    int runs;         // ceil(ow / 8)
    int act;
    long items;       // planes_out * strips * runs
    unsigned x_bytes, y_bytes; // ranges of the buffer descriptors
    // floor(2^32 / d) of the divisors of the item decode (udivmod_m)
    unsigned runs_m, strips_m, f_m, mult_m;
};

// FAST: the launcher has proved kDwSlack readable bytes in front of and behind the input (true for every tensor of a runtime arena) and
// the activation is none / ReLU: the buffer descriptor then starts kDwSlack bytes early and ends kDwSlack bytes late, so that the
// left pad of the tensor's first row is a small POSITIVE offset and a misaligned dword straddling the tensor's end is read whole (what
// it drags in is masked like any column outside the row). !FAST keeps the tensor-exact descriptor with a column-by-column path for
// the windows on the tensor's first / last row, and the transcendental activations: ~4 x the code (5.6 k instructions, 17 % of the
// time of the FAST form's loop was instruction fetch).
constexpr int kDwSlack = 64;
template <typename Tr, int R, int S, int SW, bool FAST>
__global__ __launch_bounds__(256) void conv_dw_kernel(DwArgs p) {
    constexpr int WIN = 8 * SW + S - 1; // columns of one window row
    constexpr int NL = (WIN + 7) / 8;   // 16-byte loads per window row
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    if (item >= p.items)
        return;
    unsigned q1, run_u, pl_u, strip_u, img_u, fi_u, ci_u, rem_u;
    udivmod_m((unsigned)item, (unsigned)p.runs, p.runs_m, q1, run_u);
    udivmod_m(q1, (unsigned)p.strips, p.strips_m, pl_u, strip_u);
    udivmod_m(pl_u, (unsigned)p.f, p.f_m, img_u, fi_u);
    udivmod_m(fi_u, (unsigned)p.mult, p.mult_m, ci_u, rem_u);
    const int run = (int)run_u, pl = (int)pl_u, fi = (int)fi_u;
    const int oy0 = (int)strip_u * p.th;
    const int rows = min(p.th, p.oh - oy0);
    const int x0 = run * (8 * SW) - p.pw; // input column of window column 0
    const int iy0 = oy0 * p.sh - p.ph;    // input row of ring row 0
    const __amdgpu_buffer_rsrc_t xr = FAST ? __builtin_amdgcn_make_buffer_rsrc((char *)const_cast<void *>(p.x) - kDwSlack, 0, (int)p.x_bytes + 2 * kDwSlack, 0x00020000)
                                           : __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);
    const long plane_in = ((long)img_u * p.c + ci_u) * p.h * p.wd; // element index of the input plane

    // column masks of the NL * 4 dwords of a window row (two 16-bit columns each)
    unsigned cm[NL * 4];
#pragma unroll
    for (int d = 0; d < NL * 4; ++d) {
        const unsigned lo = (unsigned)(x0 + 2 * d) < (unsigned)p.wd ? 0xffffu : 0u;
        const unsigned hi = (unsigned)(x0 + 2 * d + 1) < (unsigned)p.wd ? 0xffff0000u : 0u;
        cm[d] = lo | hi;
    }
    float wv[R * S];
    {
        const unsigned short *Wt = (const unsigned short *)p.w + (long)fi * (R * S);
#pragma unroll
        for (int k = 0; k < R * S; ++k)
            wv[k] = Tr::to_f32(Wt[k]);
    }
    const float bv = p.bias ? Tr::to_f32(((const unsigned short *)p.bias)[fi]) : 0.f;

    float ring[R][NL * 8];
    // issue: request input row iy0 + i (raw, packed) into raw[k]; commit: mask + convert raw[k] into ring slot `slot` (static).
    // The rows of output row j + 1 are requested BEFORE the arithmetic of row j (a thread has ONE window row in flight otherwise and
    // the whole chip ~3 waves per SIMD: every row step then waits out a full HBM round trip).
    u32x4_t raw[SW][NL];
    auto issue = [&](int i, auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        const int iy = iy0 + i;
        const bool rowok = iy >= 0 && iy < p.h;
        const long off = (plane_in + (long)iy * p.wd + x0) * 2 + (FAST ? kDwSlack : 0);
#pragma unroll
        for (int l = 0; l < NL; ++l)
            raw[k][l] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(xr, (rowok && off >= 0) ? (int)(off + 16 * l) : (int)0x7ffffff0, 0, 0));
        if constexpr (!FAST) {
            // (the range check works per — here misaligned — dword: a dword that straddles the END of the tensor is zeroed together
            // with the live column in its low half, and a NEGATIVE offset (the left pad of the tensor's very first row) wraps around as
            // an unsigned one. Only windows on the tensor's first and last rows can do either: those lanes re-read column by column.)
            if (rowok && (off < 0 || off + 16 * NL > (long)p.x_bytes) && off + 16 * NL > 0 && off < (long)p.x_bytes) {
#pragma unroll
                for (int e = 0; e < NL * 8; ++e) {
                    const long oe = off + 2 * e;
                    const unsigned u = (oe >= 0 && oe < (long)p.x_bytes) ? (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(xr, (int)oe, 0, 0) : 0u;
                    if ((e & 1) == 0) raw[k][e >> 3][(e >> 1) & 3] = u;
                    else raw[k][e >> 3][(e >> 1) & 3] |= u << 16;
                }
            }
        }
    };
    auto commit = [&](auto kc, auto slotc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value, slot = decltype(slotc)::value;
#pragma unroll
        for (int d = 0; d < NL * 4; ++d) {
            const unsigned v = raw[k][d >> 2][d & 3] & cm[d];
            ring[slot][2 * d] = Tr::to_f32((unsigned short)(v & 0xffffu));
            ring[slot][2 * d + 1] = Tr::to_f32((unsigned short)(v >> 16));
        }
    };
    using K0 = std::integral_constant<int, 0>;
    // ring rows 0 .. R - sh - 1, one at a time (prologue), then the request for output row 0's last sh rows
    sfor<R - SW>([&](auto ic) {
        issue(decltype(ic)::value, K0{});
        commit(K0{}, ic);
    });
    sfor<SW>([&](auto dc) { issue(R - SW + decltype(dc)::value, dc); });

    const bool relu = p.act == 1;
    for (int j0 = 0; j0 < rows; j0 += R) {
        sfor<R>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            const int j = j0 + u; // output row (relative); its window rows are ring rows j * sh + r, slot (u * sh + r) % R
            // the sh new rows of this step (input rows j * sh + R - sh + d) were requested one step ago
            sfor<SW>([&](auto dc) {
                constexpr int d = decltype(dc)::value;
                commit(dc, std::integral_constant<int, (u * SW + R - SW + d) % R>{});
            });
            // request the next step's rows (rows past the strip are harmless: in range or zeros)
            sfor<SW>([&](auto dc) { issue((j + 1) * SW + R - SW + decltype(dc)::value, dc); });
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                acc[e] = 0.f;
            sfor<R>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                constexpr int slot = (u * SW + r) % R;
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        acc[e] = fmaf(wv[r * S + s], ring[slot][e * SW + s], acc[e]);
            });
            if (j < rows) {
                const int oy = oy0 + j, ox0 = run * 8;
                unsigned short o16[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = acc[e] + bv;
                    if constexpr (FAST) v = relu ? fmaxf(v, 0.f) : v;
                    else v = relu ? fmaxf(v, 0.f) : (p.act ? apply_act(v, p.act) : v);
                    o16[e] = Tr::from_f32(v);
                }
                const long oidx = ((long)pl * p.oh + oy) * p.ow + ox0;
                if (ox0 + 8 <= p.ow) {
                    u32x4_t o;
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        o[d] = (unsigned)o16[2 * d] | ((unsigned)o16[2 * d + 1] << 16);
                    // (one 16-byte store at whatever 2-byte alignment the row start has: the buffer path takes the misaligned address
                    // like the loads do; eight 2-byte stores per run made the layers with odd row lengths store-issue-bound)
                    __builtin_amdgcn_raw_buffer_store_b128(o, yr, (int)(oidx * 2), 0, 0);
                } else {
                    unsigned short *dst = (unsigned short *)p.y + oidx;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (ox0 + e < p.ow)
                            dst[e] = o16[e];
                }
            }
        });
    }
}

template <typename Tr, int R, int S> static int launch_dw(infiniRocmRuntime_t rt, const DwArgs &p, bool fast) {
    const unsigned grid = (unsigned)ceil_div(p.items, 256);
    if (p.sw == 1 && fast)
        hipLaunchKernelGGL((conv_dw_kernel<Tr, R, S, 1, true>), dim3(grid), dim3(256), 0, rt->stream, p);
    else if (p.sw == 1)
        hipLaunchKernelGGL((conv_dw_kernel<Tr, R, S, 1, false>), dim3(grid), dim3(256), 0, rt->stream, p);
    else if (fast)
        hipLaunchKernelGGL((conv_dw_kernel<Tr, R, S, 2, true>), dim3(grid), dim3(256), 0, rt->stream, p);
    else
        hipLaunchKernelGGL((conv_dw_kernel<Tr, R, S, 2, false>), dim3(grid), dim3(256), 0, rt->stream, p);
    IROCM_LAUNCH_CHECK("conv_dw");
    return INFINI_ROCM_OK;
}

// Returns -1 when the layer does not qualify (the caller takes the generic kernel), a status otherwise.
int launch_conv_depthwise(infiniRocmRuntime_t rt, int dtype, const void *x, const void *w, const void *bias, void *y, int64_t n, int64_t c,
                          int64_t h, int64_t wd, int64_t f, int r, int s, int ph, int pw, int sh, int sw, int oh, int ow, int act) {
    if (!((r == 3 && s == 3) || (r == 5 && s == 5)) || !(sh == 1 || sh == 2) || sh != sw || f % c != 0)
        return -1;
    if (ph < 0 || pw < 0 || (((uintptr_t)x) & 1) || (((uintptr_t)y) & 1))
        return -1;
    if (n * c * h * wd * 2 >= (1ll << 31) - 256 || n * f * oh * ow * 2 >= (1ll << 31) - 64)
        return -1;
    DwArgs p;
    p.x = x; p.w = w; p.bias = bias; p.y = y;
    p.planes_out = (int)(n * f);
    p.c = (int)c; p.f = (int)f; p.mult = (int)(f / c);
    p.h = (int)h; p.wd = (int)wd; p.oh = oh; p.ow = ow;
    p.ph = ph; p.pw = pw; p.sh = sh; p.sw = sw;
    p.act = act;
    p.x_bytes = (unsigned)(n * c * h * wd * 2);
    p.y_bytes = (unsigned)(n * f * oh * ow * 2);
    p.runs = (int)ceil_div(ow, 8);
    // Rows per thread: enough threads for ~8 waves per SIMD (~512 k), strips of >= 2 R output rows so that the R - sh rows of vertical
    // halo a strip re-reads (L2 hits: the neighbouring strip runs 19 items away) and the per-thread set-up stay small. Measured
    // (batch 32, f16, rows per thread 6 / 15 / 30 / 75): C32 150 x 150 3 x 3: 31.8 / 27.9 / 39.9 / 76.9 us; C192 75 x 75 5 x 5 / 2:
    // 31.9 / 40.1 / 56.6 / 75.1 — the kernel wants threads, not long strips. Ablation of the 32 us of the first layer (loads / stores /
    // FMAs compiled out): skeleton 10 us (item decode, masks, weights, conversions, loop), FMAs 2-4, loads 8, stores 9.5 — the parts ADD
    // (one window row in flight per thread beside the prefetch), which is why it sits at 0.36-0.46 of the HBM peak on the large planes
    // and 0.15-0.25 on the 10 x 10 / 19 x 19 ones (two or three runs per row, 20-37 % of the lanes' columns dead).
    const long cols = (long)p.planes_out * p.runs;
    long strips = std::max<long>(1, std::min<long>(ceil_div(512 * 1024, cols), ceil_div(oh, 2 * r)));
    p.th = (int)ceil_div(oh, strips);
    if (const char *e = getenv("IROCM_DW_TH")) // measurement hook (tools/dwconv_bench.py --th): output rows per thread
        if (atoi(e) > 0)
            p.th = atoi(e);
    p.th = (int)ceil_div(p.th, r) * r; // (the row loop is unrolled R times)
    p.strips = (int)ceil_div(oh, p.th);
    p.items = cols * p.strips;
    if (p.items >= (1ll << 31))
        return -1;
    p.runs_m = udiv_magic((unsigned long long)p.runs);
    p.strips_m = udiv_magic((unsigned long long)p.strips);
    p.f_m = udiv_magic((unsigned long long)p.f);
    p.mult_m = udiv_magic((unsigned long long)p.mult);
    // the FAST form reads up to kDwSlack bytes in front of and behind the input through its widened descriptor: provable?
    bool fast = act == 0 || act == 1;
    if (fast) {
        hipDeviceptr_t base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)x) != hipSuccess) {
            (void)hipGetLastError();
            fast = false;
        } else {
            fast = (const char *)x - kDwSlack >= (const char *)base && (const char *)x + p.x_bytes + kDwSlack <= (const char *)base + size;
        }
    }
    if (getenv("IROCM_DW_SAFE")) // test hook (read per call): the tensor-exact form
        fast = false;
    rt->last_conv_route = "depthwise";
    if (dtype == INFINI_DT_BF16)
        return r == 3 ? launch_dw<Bf16Traits, 3, 3>(rt, p, fast) : launch_dw<Bf16Traits, 5, 5>(rt, p, fast);
    return r == 3 ? launch_dw<F16Traits, 3, 3>(rt, p, fast) : launch_dw<F16Traits, 5, 5>(rt, p, fast);
}

} // namespace irocm
