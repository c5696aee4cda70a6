// Depthwise Conv2d (groups == C, one input channel per filter; filters per channel = F / C >= 1) for f16 / bf16, NCHW x F1RS.
//
// Reference semantics: src/operators/conv.cc:47-114 (groups = C / channel_per_group; a depthwise layer is the case
// channel_per_group == 1), kernel src/kernels/cuda/conv.cc:57-168 (cuDNN group convolution), index math as the native CPU kernel
// src/kernels/cpu/conv.cc:25-50. EfficientNet-Lite4 — one of the CNNs the reference lists as validated
// (docs/SUPPORT_MATRIX_CN.md:24-27) — is built from 3 x 3 and 5 x 5 depthwise layers of stride 1 and 2. Until round 5 they fell
// through every fast route (all of which require groups == 1) to the generic implicit GEMM per (image, group) with K = R S = 9 / 25:
// a 128 x 128 x 32 MFMA tile for ONE filter row.
//
// A depthwise layer has no contraction to speak of (2 R S FLOP per output element, 18-50): it is bound by moving the input and the
// output once through HBM. Kernel: a workgroup takes PP planes x TH output rows x the whole width; it stages the input rows those
// outputs reach — converted to fp32 once, zero padding materialised (left / right pad columns and the rows above / below the image
// are zeros in LDS) — and every thread then computes runs of 8 adjacent outputs of one row from aligned 16-byte LDS reads: row r of
// the window is floats [8 k sw, 8 k sw + 8 sw + S - 1) of the staged row (the staged row starts pw floats left of column 0, so the
// window of run k starts at a multiple of 8 floats). R S weights of the channel live in registers; fp32 accumulation in the oracle's
// tap order (r outer, s inner); bias + activation fused; 16-byte stores (element stores for the ragged last run of a row).
#include "gemm_common.h"

namespace irocm {

struct DwArgs {
    const void *x, *w, *bias;
    void *y;
    int planes_out;   // N * F
    int c, f, mult;   // mult = F / C filters per input channel: filter fi reads channel fi / mult
    int h, wd, oh, ow;
    int ph, pw, sh, sw;
    int th, pp;       // output rows and planes per workgroup
    int strips;       // ceil(oh / th)
    int ih;           // staged input rows per plane: (th - 1) * sh + R
    int pitch;        // floats per staged row (multiple of 4, >= the widest window)
    int runs;         // ceil(ow / 8)
    int act;
    unsigned x_bytes, y_bytes; // ranges of the buffer descriptors
    // floor(2^32 / d) of the divisors of the item decode (udivmod_m: a division by a run-time value is ~45 instructions, the decode
    // of an item had five of them — more than the item's arithmetic)
    unsigned chunks_m, ih_m, f_m, mult_m, runs_m, th_m;
};

template <typename Tr, int R, int S, int SW>
__global__ __launch_bounds__(256) void conv_dw_kernel(DwArgs p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int t = threadIdx.x;
    const int strip = blockIdx.x % p.strips;
    const int pg = blockIdx.x / p.strips; // plane group
    const int plane0 = pg * p.pp;
    const int npl = min(p.pp, p.planes_out - plane0);
    const int oy0 = strip * p.th;
    const int rows = min(p.th, p.oh - oy0);
    const int iy0 = oy0 * p.sh - p.ph; // input row of staged row 0
    const int ih = p.ih; // (a short last strip stages the full strip's rows: what lies below the image is zeros)
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);

    // ---- stage: npl planes x ih rows x pitch floats; staged column j holds input column j - pw ------------------------------
    // work items: (plane, row, 8-float chunk of the row); a chunk whose 8 columns all lie inside the image row is one 16-byte load
    // (2-byte aligned: image rows start anywhere), anything else element by element (row ends, pads, rows outside the image)
    const int chunks = p.pitch / 8;
    const int items = npl * ih * chunks;
    for (int it = t; it < items; it += 256) {
        unsigned rowi_u, ch_u, lpl_u, ri_u, img_u, fi_u, ci_u, rem_u;
        udivmod_m((unsigned)it, (unsigned)chunks, p.chunks_m, rowi_u, ch_u); // plane-major staged row index, chunk
        udivmod_m(rowi_u, (unsigned)ih, p.ih_m, lpl_u, ri_u);
        const int ch = (int)ch_u, rowi = (int)rowi_u;
        const int iy = iy0 + (int)ri_u;
        const int pl = plane0 + (int)lpl_u;
        udivmod_m((unsigned)pl, (unsigned)p.f, p.f_m, img_u, fi_u);
        udivmod_m(fi_u, (unsigned)p.mult, p.mult_m, ci_u, rem_u);
        const int img = (int)img_u, ci = (int)ci_u;
        const long rowbase = (((long)img * p.c + ci) * p.h + iy) * p.wd; // element index of (row iy, column 0)
        const int x0 = ch * 8 - p.pw;                                    // input column of the chunk's first float
        // Branch-free: ONE 16-byte load per chunk wherever it lies (2-byte aligned; bytes outside the tensor read as zeros through the
        // descriptor's range check), then the elements outside the image row are replaced by zeros with selects. (The first version
        // took row ends and pads element by element under per-lane conditions: every wave has such lanes, so every wave ran eight
        // predicated 2-byte loads with a wait behind each — the PMC pass showed 61 % of the wave cycles waiting and 0.56 SALU
        // instructions per VALU.)
        const bool rowok = iy >= 0 && iy < p.h;
        const long off = (rowbase + x0) * 2;
        u32x4_t q = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(xr, (rowok && off >= 0) ? (int)off : (int)0x7ffffff0, 0, 0));
        // (the range check works per — here misaligned — dword: a dword that straddles the END of the tensor is zeroed together with the
        // live element in its low half; and a NEGATIVE offset (the left pad of the tensor's very first row) wraps around as an unsigned
        // one. Only chunks of the tensor's first and last row can do either: those lanes re-read element by element; a wave without
        // such a lane skips the branch.)
        if (rowok && (off < 0 || off + 16 > (long)p.x_bytes) && off + 16 > 0 && off < (long)p.x_bytes) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const long oe = off + 2 * e;
                const unsigned u = (oe >= 0 && oe < (long)p.x_bytes) ? (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(xr, (int)oe, 0, 0) : 0u;
                if ((e & 1) == 0) q[e >> 1] = u;
                else q[e >> 1] |= u << 16;
            }
        }
        float v[8];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            v[2 * d] = Tr::to_f32((unsigned short)(q[d] & 0xffffu));
            v[2 * d + 1] = Tr::to_f32((unsigned short)(q[d] >> 16));
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = (rowok && (unsigned)(x0 + e) < (unsigned)p.wd) ? v[e] : 0.f;
        float4 *dst = (float4 *)(sm + (long)rowi * p.pitch + ch * 8);
        dst[0] = make_float4(v[0], v[1], v[2], v[3]);
        dst[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
    __syncthreads();

    // ---- compute: items (plane, output row, run of 8 outputs) --------------------------------------------------------------------
    constexpr int WIN = 8 * SW + S - 1;       // floats of one window row
    constexpr int NV = (WIN + 3) / 4;         // 16-byte reads per window row
    const int citems = npl * rows * p.runs;
    const unsigned short *Wt = (const unsigned short *)p.w;
    const unsigned short *bias = (const unsigned short *)p.bias;
    unsigned short *Y = (unsigned short *)p.y;
    for (int it = t; it < citems; it += 256) {
        unsigned rr_u, run_u, lp_u, ty_u, img_u, fi_u;
        udivmod_m((unsigned)it, (unsigned)p.runs, p.runs_m, rr_u, run_u);
        if (rows == p.th) {
            udivmod_m(rr_u, (unsigned)p.th, p.th_m, lp_u, ty_u);
        } else { // (the last strip of a plane: pp == 1 whenever strips > 1)
            lp_u = rr_u / (unsigned)rows;
            ty_u = rr_u - lp_u * (unsigned)rows;
        }
        const int run = (int)run_u, ty = (int)ty_u, lp = (int)lp_u; // lp: local plane
        const int pl = plane0 + lp;
        udivmod_m((unsigned)pl, (unsigned)p.f, p.f_m, img_u, fi_u);
        const int fi = (int)fi_u;
        float wv[R * S];
#pragma unroll
        for (int k = 0; k < R * S; ++k)
            wv[k] = Tr::to_f32(Wt[(long)fi * (R * S) + k]);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            acc[e] = 0.f;
        const float *srow = sm + ((long)lp * ih + ty * p.sh) * p.pitch + run * (8 * SW);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float win[NV * 4];
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const float4 vv = ((const float4 *)(srow + (long)r * p.pitch))[q];
                win[4 * q] = vv.x; win[4 * q + 1] = vv.y; win[4 * q + 2] = vv.z; win[4 * q + 3] = vv.w;
            }
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    acc[e] = fmaf(wv[r * S + s], win[e * SW + s], acc[e]);
        }
        const float bv = bias ? Tr::to_f32(bias[fi]) : 0.f;
        const int oy = oy0 + ty, ox0 = run * 8;
        unsigned short *dst = Y + ((long)pl * p.oh + oy) * p.ow + ox0;
        unsigned short o16[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            o16[e] = Tr::from_f32(apply_act(acc[e] + bv, p.act));
        if (ox0 + 8 <= p.ow) {
            u32x4_t o;
#pragma unroll
            for (int d = 0; d < 4; ++d)
                o[d] = (unsigned)o16[2 * d] | ((unsigned)o16[2 * d + 1] << 16);
            // (one 16-byte store at whatever 2-byte alignment the row start has: rows of odd length start on odd elements, and eight
            // 2-byte stores per run made those layers store-issue-bound; the buffer path takes the misaligned address like the loads do)
            __builtin_amdgcn_raw_buffer_store_b128(o, yr, (int)((((long)pl * p.oh + oy) * p.ow + ox0) * 2), 0, 0);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (ox0 + e < p.ow)
                    dst[e] = o16[e];
        }
    }
}

template <typename Tr, int R, int S> static int launch_dw(infiniRocmRuntime_t rt, const DwArgs &p, size_t lds) {
    const unsigned grid = (unsigned)(ceil_div(p.planes_out, p.pp) * p.strips);
#define IROCM_DW(SWV)                                                       \
    do {                                                                    \
        auto kern = conv_dw_kernel<Tr, R, S, SWV>;                          \
        /* (<= 48 KiB of dynamic LDS: below the 64 KiB every kernel may ask for without an attribute) */ \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, rt->stream, p); \
    } while (0)
    if (p.sw == 1) IROCM_DW(1);
    else IROCM_DW(2);
#undef IROCM_DW
    IROCM_LAUNCH_CHECK("conv_dw");
    return INFINI_ROCM_OK;
}

// Returns -1 when the layer does not qualify (the caller takes the generic kernel), a status otherwise.
int launch_conv_depthwise(infiniRocmRuntime_t rt, int dtype, const void *x, const void *w, const void *bias, void *y, int64_t n, int64_t c,
                          int64_t h, int64_t wd, int64_t f, int r, int s, int ph, int pw, int sh, int sw, int oh, int ow, int act) {
    if (!((r == 3 && s == 3) || (r == 5 && s == 5)) || !(sh == 1 || sh == 2) || sh != sw || f % c != 0)
        return -1;
    if (ph < 0 || pw < 0 || pw > 8 || (((uintptr_t)x) & 1) || (((uintptr_t)y) & 1))
        return -1;
    if (n * c * h * wd * 2 >= (1ll << 31) - 64 || n * f * oh * ow * 2 >= (1ll << 31) - 64)
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
    // staged row: columns -pw .. ; the widest window is run (runs - 1): floats up to (runs - 1) * 8 sw + 8 sw + S - 1, rounded to a chunk
    p.pitch = (int)ceil_div((long)p.runs * 8 * sw + s - 1, 8) * 8;
    // ~2 work items (runs of 8 outputs) per thread and workgroup: whole planes while they are small, row strips otherwise; LDS <= 48 KiB
    const long want = 512;
    const long per_plane = (long)p.runs * oh;
    const long lds_cap = 48 * 1024 / 4; // floats
    if (per_plane >= want) {
        p.pp = 1;
        p.th = (int)std::max<long>(1, std::min<long>(oh, ceil_div(want, p.runs)));
    } else {
        p.th = oh;
        p.pp = (int)std::max<long>(1, std::min<long>(p.planes_out, want / per_plane));
    }
    auto ih_of = [&](int th) { return (th - 1) * sh + r; };
    while (p.pp > 1 && (long)p.pp * ih_of(p.th) * p.pitch > lds_cap)
        --p.pp;
    while (p.th > 1 && (long)p.pp * ih_of(p.th) * p.pitch > lds_cap)
        --p.th;
    if ((long)p.pp * ih_of(p.th) * p.pitch > lds_cap)
        return -1; // one output row's window does not fit (rows of > ~1500 pixels): generic kernel
    p.ih = ih_of(p.th);
    p.strips = (int)ceil_div(oh, p.th);
    if (ceil_div(p.planes_out, p.pp) * p.strips >= (1ll << 31))
        return -1;
    p.chunks_m = udiv_magic((unsigned long long)(p.pitch / 8));
    p.ih_m = udiv_magic((unsigned long long)p.ih);
    p.f_m = udiv_magic((unsigned long long)p.f);
    p.mult_m = udiv_magic((unsigned long long)p.mult);
    p.runs_m = udiv_magic((unsigned long long)p.runs);
    p.th_m = udiv_magic((unsigned long long)p.th);
    const size_t lds = (size_t)p.pp * p.ih * p.pitch * sizeof(float);
    rt->last_conv_route = "depthwise";
    if (dtype == INFINI_DT_BF16)
        return r == 3 ? launch_dw<Bf16Traits, 3, 3>(rt, p, lds) : launch_dw<Bf16Traits, 5, 5>(rt, p, lds);
    return r == 3 ? launch_dw<F16Traits, 3, 3>(rt, p, lds) : launch_dw<F16Traits, 5, 5>(rt, p, lds);
}

} // namespace irocm
