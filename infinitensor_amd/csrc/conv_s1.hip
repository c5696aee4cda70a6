// conv_s1: the fast Conv2d path for gfx950 — unit-stride "same" convolutions (OH == H, OW == W,
// dilation 1, groups 1, C % BK == 0) in f16 / bf16 as an implicit GEMM on MFMA, NCHW in, NCHW out,
// no im2col buffer and no layout change of the activations in HBM.
//
// Replaces convCudnn for these shapes (reference: src/kernels/cuda/conv.cc:57-168); semantics as the
// native CPU kernel src/kernels/cpu/conv.cc:25-50. ResNet-50's 3x3 and (small-plane) 1x1 layers land here.
//
// GEMM view:  Y[f, col] = sum_{rs} sum_{c} W'[rs][f][c] * X[img(col), c, pix(col) + (r - ph) * W + (s - pw)]
//   * columns are "pixel slots": col = img * HWp + pix, HWp = H*W rounded up to 8, so a 16-byte run of 8
//     slots never crosses an image and tiles span images (a 7x7 or 14x14 plane does not waste a tile).
//     For a same-size unit-stride convolution the 8 input elements of a run are CONTIGUOUS in memory for
//     every tap (r, s) — the tap only shifts the address — so the B operand is fetched with one 16-byte
//     buffer load per run (2-byte aligned: measured on MI355X at 4.5 TB/s vs 6.4 TB/s aligned, tools/probes/
//     bufprobe.hip) and the halo / zero padding is an 8-bit mask per (run, tap) built once per thread from
//     row / column validity bit sets. Runs that would touch bytes outside the tensor are fetched by element.
//   * K order is tap-major (rs outer, channel inner): the masks and the address shift change only every C/BK
//     K-steps. Weights are re-packed once per call FCRS -> [RS][F][C] (tiny; K-major 16-byte runs).
//   * tile BM x BN x BK with 4 waves of 64x64 (4 x 4 MFMA 16x16x32, swapped operands so a lane owns 4
//     consecutive pixel slots of one filter): <2,2,64> = 128 x 128 x 64, <1,4,32> = 64 x 256 x 32 for F <= 64.
//     A -> LDS [BM][BK + 8] (ds_read_b128), B -> LDS [BK][BN] XOR-swizzled, read with ds_read_b64_tr_b16.
//     Register-staged double buffering: the global loads of K-step t+1 are in flight during the MFMAs of
//     step t; one barrier per K-step; 2-3 workgroups per CU hide each other's barriers.
//   * epilogue: + bias[f], activation, 8-byte stores of 4 pixels when the plane size allows.
//   * ROWTAP variant for channel counts that are not a multiple of 32 (the 3-channel 7x7/2 stem): K is the flat
//     index k = tap * C + c padded to 32, weights are re-packed to [F][Kpad], and every B row carries its own
//     (plane, shift, mask) — decoded per row per K-step, which is cheap next to the 16 MFMAs it feeds.
//   * strides > 1 (the ResNet down-sampling layers): a pre-pass de-interleaves X into the sh*sw "phase planes"
//     Xp[py][px][n][c][OH][OW] = X[n][c][i*sh + py][j*sw + px] that some tap reads (1 of 4 for a 1x1/2, all 4 for a
//     3x3/2); on a phase plane every tap is again a constant shift of a unit-stride same-size access, so the same
//     kernel runs with a per-tap (plane, shift) pair. Needs OH == ceil(H / sh) and OW == ceil(W / sw).
#include "gemm256_common.h"
#include <type_traits>
#include <cstdlib>

namespace irocm {
// gemm256p_conv.hip
int launch_conv_pw_gemm(infiniRocmRuntime_t rt, int dtype, const void *x, const void *w, const void *bias, const void *res, void *y,
                        int64_t n, int64_t c, int64_t hw, int64_t f, int act);
} // namespace irocm

namespace irocm {

int launch_conv_tap_gemm(infiniRocmRuntime_t rt, int dtype, const void *x, const void *wp, const void *bias, void *y, int64_t n,
                         int64_t c, int oh, int ow, int in_h, int in_w, int stride, int64_t plane_elems, int64_t f, int act,
                         int split, void *slab, size_t slab_bytes);
int conv_tap_split(infiniRocmRuntime_t rt, int64_t n, int64_t hw, int64_t c, int64_t f, size_t *slab_bytes);

template <int N> __device__ __forceinline__ void g256p_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct ConvS1Args {
    const void *x, *w, *bias; // x: input or its phase planes; w: [RS][F][C]
    const void *res;          // optional residual, same shape as y, added before the activation
    void *y;
    int nimg, c, f, r, s, ph, pw, sh, sw, dh, dw;
    int in_h, in_w;  // input extent (padding validity)
    int h, wd;       // plane extent = output extent
    int hw, hwp, ncols;
    int tiles_m, tiles_n;
    int act;
    int kdim, kpad;      // ROWTAP: C*R*S and its round-up to 32 (w is [F][kpad])
    int wide_epilogue;   // LDS-staged 16-byte-run epilogue: 0 off, 1 even planes only, 2 also odd planes (IROCM_CONV_WIDE)
    int epi_probe;       // ablation hook, 0 in production
    unsigned hwp_m, wd_m; // floor(2^32 / hwp), floor(2^32 / wd): divisions by multiply-high + one correction (fast_divmod)
    unsigned x_bytes;    // bytes of everything behind x
    unsigned y_bytes;    // bytes of y (= bytes of the residual); 0 when they do not fit 32-bit buffer offsets
    long plane_elems;    // elements of one phase plane set [n][c][h][wd]
    signed char slot[16]; // phase py*sw + px -> index of its plane set behind x
};

// n / d and n % d for 0 <= n < 2^32 with m = min(floor(2^32 / d), 2^32 - 1): the multiply-high quotient is short by at most one.
// (An integer division by a run-time value is ~45 VALU instructions; the conv prologues did 8-16 of them per lane.)
__device__ __forceinline__ void fast_divmod(int n, int d, unsigned m, int &q, int &r) {
    q = (int)__umulhi((unsigned)n, m);
    r = n - q * d;
    if (r >= d) {
        ++q;
        r -= d;
    }
}
static inline unsigned divmod_magic(long d) { return d <= 1 ? 0xffffffffu : (unsigned)(((1ull << 32)) / (unsigned long long)d); }

struct PhaseSplitArgs {
    const unsigned short *x;
    unsigned short *o;
    long planes; // n * c
    int in_h, in_w, oh, ow, sh, sw, nslots;
    signed char py[16], px[16];
};

// o[slot][plane][i][j] = x[plane][i*sh + py][j*sw + px] (0 outside the input)
__global__ __launch_bounds__(256) void conv_phase_split(PhaseSplitArgs a) {
    const long per_slot = a.planes * a.oh * a.ow;
    const long total = per_slot * a.nslots;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int sl = (int)(i / per_slot);
        long q = i - (long)sl * per_slot;
        const int j = (int)(q % a.ow);
        q /= a.ow;
        const int ii = (int)(q % a.oh);
        const long pl = q / a.oh;
        const int ih = ii * a.sh + a.py[sl], iw = j * a.sw + a.px[sl];
        a.o[i] = (ih < a.in_h && iw < a.in_w) ? a.x[(pl * a.in_h + ih) * a.in_w + iw] : (unsigned short)0;
    }
}

// 1 x 1 / 2 layers read ONE phase: o[plane][oy][ox] = x[plane][2 oy][2 ox]. One thread takes VEC consecutive input columns of an
// even input row (one VEC * 2-byte load) and stores the VEC / 2 even ones; only the rows and columns that are read are touched
// (the generic kernel spent three emulated divisions and a 2-byte load per OUTPUT element: 30 us for 6-13 MB of output).
struct Subsample2Args {
    const unsigned short *x;
    unsigned short *o;
    int planes, in_h, in_w, oh, ow;
    unsigned groups_m, oh_m; // multiply-high reciprocals of the column groups per row and of oh
};
template <int VEC> __global__ __launch_bounds__(256) void conv_subsample2_kernel(Subsample2Args a) {
    const int groups = a.in_w / VEC; // in_w % VEC == 0 (launcher)
    const int total = a.planes * a.oh * groups;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int q, g, pl, oy;
        fast_divmod(i, groups, a.groups_m, q, g);
        fast_divmod(q, a.oh, a.oh_m, pl, oy);
        const unsigned short *src = a.x + ((long)pl * a.in_h + 2 * oy) * a.in_w + VEC * g;
        unsigned short *dst = a.o + ((long)pl * a.oh + oy) * a.ow + (VEC / 2) * g;
        if constexpr (VEC == 8) {
            const u32x4_t v = *(const u32x4_t *)src;
            u32x2_t ev;
            ev[0] = (v[0] & 0xffffu) | (v[1] << 16);
            ev[1] = (v[2] & 0xffffu) | (v[3] << 16);
            *(u32x2_t *)dst = ev;
        } else if constexpr (VEC == 4) {
            const u32x2_t v = *(const u32x2_t *)src;
            *(unsigned *)dst = (v[0] & 0xffffu) | (v[1] << 16);
        } else {
            *dst = (unsigned short)(*(const unsigned *)src & 0xffffu);
        }
    }
}

__global__ __launch_bounds__(256) void conv_repack_w(const unsigned short *__restrict__ w,
                                                     unsigned short *__restrict__ o, int f, int c, int rs) {
    // o[t][f][c] = w[f][c][t]
    const long total = (long)f * c * rs;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cc = (int)(i % c);
        const long q = i / c;
        const int ff = (int)(q % f);
        const int t = (int)(q / f);
        o[i] = w[((long)ff * c + cc) * rs + t];
    }
}

// Stride 2 x 2 (every ResNet down-sampling layer): one thread per 4 consecutive input columns of one (virtual) input
// row -> two outputs in each of the two column phases of that row's phase; 32-bit index math.
// slot[py*2 + px] < 0: that phase is not read by any tap and is not materialised.
struct PhaseSplit2Args {
    const unsigned short *x;
    unsigned short *o;
    int planes, in_h, in_w, oh, ow;
    long plane_elems; // planes * oh * ow
    signed char slot[4];
};
__global__ __launch_bounds__(256) void conv_phase_split_2x2(PhaseSplit2Args a) {
    const int quads = (a.ow + 1) / 2; // 4 input columns = 2 output columns per thread
    const int rows = 2 * a.oh;
    const int total = a.planes * rows * quads;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int jq = i % quads;
        const int q = i / quads;
        const int iy = q % rows, pl = q / rows;
        const int py = iy & 1, oy = iy >> 1;
        if (a.slot[py * 2] < 0 && a.slot[py * 2 + 1] < 0)
            continue; // no tap reads this row parity (1x1 / 2: odd rows)
        const unsigned short *src = a.x + ((long)pl * a.in_h + iy) * a.in_w + 4 * jq;
        unsigned short v[4] = {0, 0, 0, 0};
        if (iy < a.in_h) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * jq + e < a.in_w)
                    v[e] = src[e];
        }
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            const int sl = a.slot[py * 2 + px];
            if (sl < 0)
                continue;
            unsigned short *dst = a.o + sl * a.plane_elems + ((long)pl * a.oh + oy) * a.ow + 2 * jq;
            dst[0] = v[px];
            if (2 * jq + 1 < a.ow)
                dst[1] = v[2 + px];
        }
    }
}

// Same split, 8 input columns per thread: one 16-byte load, one 8-byte store into each column phase. Needs
// in_w % 8 == 0 (so ow % 4 == 0), 16-byte aligned x and 8-byte aligned planes. Rows past the input (odd heights) are
// written as zeros like in the scalar kernels.
__global__ __launch_bounds__(256) void conv_phase_split_2x2_v8(PhaseSplit2Args a) {
    const int octs = a.in_w / 8;
    const int rows = 2 * a.oh;
    const int total = a.planes * rows * octs;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int jo = i % octs;
        const int q = i / octs;
        const int iy = q % rows, pl = q / rows;
        const int py = iy & 1, oy = iy >> 1;
        if (a.slot[py * 2] < 0 && a.slot[py * 2 + 1] < 0)
            continue;
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (iy < a.in_h)
            v = *(const u32x4_t *)(a.x + ((long)pl * a.in_h + iy) * a.in_w + 8 * jo);
        // even / odd elements of the 8: dword d holds elements 2d (low half) and 2d+1 (high half)
        u32x2_t ev, od;
        ev[0] = (v[0] & 0xffffu) | (v[1] << 16);
        ev[1] = (v[2] & 0xffffu) | (v[3] << 16);
        od[0] = (v[0] >> 16) | (v[1] & 0xffff0000u);
        od[1] = (v[2] >> 16) | (v[3] & 0xffff0000u);
        const long off = ((long)pl * a.oh + oy) * a.ow + 4 * jo;
        if (a.slot[py * 2] >= 0)
            *(u32x2_t *)(a.o + a.slot[py * 2] * a.plane_elems + off) = ev;
        if (a.slot[py * 2 + 1] >= 0)
            *(u32x2_t *)(a.o + a.slot[py * 2 + 1] * a.plane_elems + off) = od;
    }
}

// The 2 x 2 split for rows of in_w % 4 == 0 (VEC = 4: one 8-byte load, one 4-byte store per column phase) or in_w % 2 == 0
// (VEC = 2: one 4-byte load, one 2-byte store per phase) columns — ResNet's 28- and 14-pixel rows, which the 8-column kernel
// cannot take — with multiply-high index math. Rows past the input (odd heights) are written as zeros.
struct PhaseSplit2vArgs {
    PhaseSplit2Args a;
    unsigned groups_m, rows_m;
};
template <int VEC> __global__ __launch_bounds__(256) void conv_phase_split_2x2_vn(PhaseSplit2vArgs v) {
    const PhaseSplit2Args &a = v.a;
    const int groups = a.in_w / VEC;
    const int rows = 2 * a.oh;
    const int total = a.planes * rows * groups;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int q, g, pl, iy;
        fast_divmod(i, groups, v.groups_m, q, g);
        fast_divmod(q, rows, v.rows_m, pl, iy);
        const int py = iy & 1, oy = iy >> 1;
        if (a.slot[py * 2] < 0 && a.slot[py * 2 + 1] < 0)
            continue;
        const unsigned short *src = a.x + ((long)pl * a.in_h + iy) * a.in_w + VEC * g;
        const long off = ((long)pl * a.oh + oy) * a.ow + (VEC / 2) * g;
        if constexpr (VEC == 4) {
            u32x2_t w = {0u, 0u};
            if (iy < a.in_h)
                w = *(const u32x2_t *)src;
            if (a.slot[py * 2] >= 0)
                *(unsigned *)(a.o + a.slot[py * 2] * a.plane_elems + off) = (w[0] & 0xffffu) | (w[1] << 16);
            if (a.slot[py * 2 + 1] >= 0)
                *(unsigned *)(a.o + a.slot[py * 2 + 1] * a.plane_elems + off) = (w[0] >> 16) | (w[1] & 0xffff0000u);
        } else {
            unsigned w = 0u;
            if (iy < a.in_h)
                w = *(const unsigned *)src;
            if (a.slot[py * 2] >= 0)
                a.o[a.slot[py * 2] * a.plane_elems + off] = (unsigned short)(w & 0xffffu);
            if (a.slot[py * 2 + 1] >= 0)
                a.o[a.slot[py * 2 + 1] * a.plane_elems + off] = (unsigned short)(w >> 16);
        }
    }
}

// o[f][t*c + cc] = w[f][cc][t], zero for k in [c*rs, kpad)
__global__ __launch_bounds__(256) void conv_repack_w_flat(const unsigned short *__restrict__ w,
                                                          unsigned short *__restrict__ o, int f, int c, int rs, int kpad) {
    const long total = (long)f * kpad;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i % kpad);
        const int ff = (int)(i / kpad);
        const int t = k / c, cc = k - t * c;
        o[i] = k < c * rs ? w[((long)ff * c + cc) * rs + t] : (unsigned short)0;
    }
}

// Epilogue of one 128 x 128 (WM = WN = 2) or 64 x 256 workgroup tile: + bias[f], optional residual, activation, NCHW store.
// The accumulators hold filters on lanes (l15) and 4 consecutive pixel slots per lane (g4 * 4 + r).
template <typename Tr>
__device__ __forceinline__ void conv_tile_epilogue(const ConvS1Args &p, f32x4 (&acc)[4][4], int m0, int n0, int wm, int wn,
                                                   int l15, int g4) {
    unsigned short *Y = (unsigned short *)p.y;
    const unsigned short *bias = (const unsigned short *)p.bias;
    if (p.wide_epilogue == 9) { // ablation hook (IROCM_CONV_WIDE=9): keep the accumulators alive, store almost nothing
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (sacc == 123.456f)
            Y[0] = 1;
        return;
    }
    const bool vec_ok = (p.hw % 4 == 0) && ((((uintptr_t)p.y) & 7) == 0);
    // Wide path (planes that are a multiple of 16 pixels, i.e. hwp == hw: 56x56, 28x28, ...): lane groups g4 / g4^1 swap
    // halves of a pair of 16-pixel tiles so that every lane stores 8 consecutive pixels of one filter row with ONE
    // 16-byte store (same exchange as the GEMM epilogue); wave-uniform condition, so the shuffles are convergent.
    const bool wide = p.wide_epilogue && (p.hw % 16 == 0) && ((((uintptr_t)p.y) & 15) == 0) && (n0 + wn * 64 + 64 <= p.ncols) &&
                      (m0 + wm * 64 + 64 <= p.f) && (!p.res || ((((uintptr_t)p.res) & 7) == 0));
    if (wide) {
        const bool odd = g4 & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int fm = m0 + wm * 64 + i * 16 + l15;
            const float bv = bias ? Tr::to_f32(bias[fm]) : 0.f;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                unsigned pk[2][2];
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    const int col = n0 + wn * 64 + (jp * 2 + t2) * 16 + g4 * 4;
                    const int im = col / p.hw, pix = col - im * p.hw; // hwp == hw here
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        v[r] = acc[i][jp * 2 + t2][r] + bv;
                    if (p.res) {
                        const u32x2_t rk = *(const u32x2_t *)((const unsigned short *)p.res + ((long)im * p.f + fm) * p.hw + pix);
                        v[0] += Tr::to_f32((unsigned short)(rk[0] & 0xffff)); v[1] += Tr::to_f32((unsigned short)(rk[0] >> 16));
                        v[2] += Tr::to_f32((unsigned short)(rk[1] & 0xffff)); v[3] += Tr::to_f32((unsigned short)(rk[1] >> 16));
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        v[r] = apply_act(v[r], p.act);
                    pk[t2][0] = Tr::pack2(v[0], v[1]);
                    pk[t2][1] = Tr::pack2(v[2], v[3]);
                }
                const unsigned s0 = odd ? pk[0][0] : pk[1][0], s1 = odd ? pk[0][1] : pk[1][1];
                const unsigned r0 = (unsigned)__shfl_xor((int)s0, 16), r1 = (unsigned)__shfl_xor((int)s1, 16);
                u32x4_t o;
                if (odd) { o[0] = r0; o[1] = r1; o[2] = pk[1][0]; o[3] = pk[1][1]; }
                else { o[0] = pk[0][0]; o[1] = pk[0][1]; o[2] = r0; o[3] = r1; }
                const int col = n0 + wn * 64 + (jp * 2 + (odd ? 1 : 0)) * 16 + (g4 & ~1) * 4;
                const int im = col / p.hw, pix = col - im * p.hw;
                *(u32x4_t *)(Y + ((long)im * p.f + fm) * p.hw + pix) = o;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 64 + j * 16 + g4 * 4;
        if (col >= p.ncols)
            continue;
        const int im = col / p.hwp, pix = col - im * p.hwp;
        if (pix >= p.hw)
            continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int fm = m0 + wm * 64 + i * 16 + l15;
            if (fm >= p.f)
                continue;
            const float bv = bias ? Tr::to_f32(bias[fm]) : 0.f;
            const long yoff = ((long)im * p.f + fm) * p.hw + pix;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                v[r] = acc[i][j][r] + bv;
            if (p.res) {
                const unsigned short *rp = (const unsigned short *)p.res + yoff;
                if (vec_ok && pix + 3 < p.hw && ((((uintptr_t)p.res) & 7) == 0)) {
                    const u32x2_t rk = *(const u32x2_t *)rp;
                    v[0] += Tr::to_f32((unsigned short)(rk[0] & 0xffff)); v[1] += Tr::to_f32((unsigned short)(rk[0] >> 16));
                    v[2] += Tr::to_f32((unsigned short)(rk[1] & 0xffff)); v[3] += Tr::to_f32((unsigned short)(rk[1] >> 16));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (pix + r < p.hw)
                            v[r] += Tr::to_f32(rp[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                v[r] = apply_act(v[r], p.act);
            unsigned short *dst = Y + yoff;
            if (vec_ok && pix + 3 < p.hw) {
                u32x2_t pk;
                pk[0] = Tr::pack2(v[0], v[1]);
                pk[1] = Tr::pack2(v[2], v[3]);
                *(u32x2_t *)dst = pk;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (pix + r < p.hw)
                        dst[r] = Tr::from_f32(v[r]);
            }
        }
    }
}

// LDS-staged epilogue: measured on the pointwise layers, the direct stores above are what the convolutions wait for
// (256->1024 @14x14: 26 us without the stores, 88 us with them; 128->512 @28x28: 23 vs 69 us) -- a wave store that
// scatters 32- or 64-byte pieces over 16 filter rows moves < 1-2 TB/s. Here every wave first writes its 64 x 64 tile
// (bias + activation applied, rounded) into a private [64][72]-element LDS image, then stores it row-wise: 8 lanes x 16
// bytes = 128 contiguous bytes per filter row, 8 rows per instruction, one integer division per lane per tile.
// `wbuf`: this wave's 9216 bytes. Needs an even plane size (dword-aligned runs).
constexpr int kEpiWaveBytes = 64 * 144;
// The bias values of the lane's four filter rows (m0 + wm*64 + i*16 + l15) are loaded by the caller EARLY (a 2-byte
// global load issued here would put an L2 round trip in front of every tile's stores).
template <typename Tr>
__device__ __forceinline__ void conv_load_bias(const ConvS1Args &p, int m0, int wm, int l15, float (&bv)[4]) {
    const unsigned short *bias = (const unsigned short *)p.bias;
    // unconditional loads (filter index clamped) under ONE uniform branch: as `cond ? bias[fm] : 0` every load sat in its own
    // branch with an s_waitcnt vmcnt(0) behind it — four serialised round trips in the kernel prologue
    if (bias) {
        unsigned short raw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int fm = m0 + wm * 64 + i * 16 + l15;
            raw[i] = bias[fm < p.f ? fm : p.f - 1];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            bv[i] = Tr::to_f32(raw[i]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            bv[i] = 0.f;
    }
}

template <typename Tr>
__device__ __forceinline__ void conv_tile_epilogue_lds(const ConvS1Args &p, f32x4 (&acc)[4][4], const float (&bv)[4], int m0,
                                                       int n0, int wm, int wn, int lane, char *wbuf) {
    constexpr int ROWP = 144;
    const int l15 = lane & 15, g4 = lane >> 4;
    // row-wise phase geometry: lane = (row sub-index, 16-byte chunk); one integer division per lane per tile
    const int ch = lane & 7, rsub = lane >> 3;
    const int col = n0 + wn * 64 + ch * 8;
    int im, pix;
    fast_divmod(col, p.hwp, p.hwp_m, im, pix);
    const bool live = col < p.ncols && pix < p.hw;
    const bool full = pix + 8 <= p.hw;
    const long ybase_off = (long)im * p.f * p.hw + pix;
    // Residual (y = act(conv + bias + residual), the join of a ResNet bottleneck): the lane's eight 16-byte runs are
    // fetched HERE, ahead of the staging below, through a range-checked buffer descriptor (a run that ends a plane of
    // hw % 8 != 0 pixels may reach past the tensor: those dwords read as zero and are never stored). They are added in
    // the row-wise phase, so the residual moves in the same 128-byte row segments as the stores.
    const bool has_res = p.res != nullptr;
    u32x4_t rv[8];
    if (has_res) {
        const __amdgpu_buffer_rsrc_t rrs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.res), 0, (int)p.y_bytes, 0x00020000);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            int fm = m0 + wm * 64 + it * 8 + rsub;
            fm = fm < p.f ? fm : p.f - 1;
            const long off = live ? (ybase_off + (long)fm * p.hw) * 2 : 0;
            rv[it] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)off, 0, 0);
        }
    }
    // one copy of the packing loop per activation: a runtime switch per element keeps 64 scalar branches in the loop
    auto stage = [&](auto actc) __attribute__((always_inline)) {
        constexpr int ACT = decltype(actc)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[i][j][r] + bv[i];
                    if constexpr (ACT == 1)
                        v[r] = v[r] > 0.f ? v[r] : 0.f;
                    else if constexpr (ACT < 0)
                        v[r] = apply_act(v[r], p.act);
                }
                u32x2_t pk;
                pk[0] = Tr::pack2(v[0], v[1]);
                pk[1] = Tr::pack2(v[2], v[3]);
                *(u32x2_t *)(wbuf + (i * 16 + l15) * ROWP + (j * 16 + g4 * 4) * 2) = pk;
            }
        }
    };
    if (p.act == 0 || has_res) // with a residual the activation follows the add, in the row-wise phase
        stage(std::integral_constant<int, 0>{});
    else if (p.act == 1)
        stage(std::integral_constant<int, 1>{});
    else
        stage(std::integral_constant<int, -1>{});
    __builtin_amdgcn_wave_barrier(); // same wave writes and reads: LDS operations of one wave complete in order
    unsigned short *ybase = (unsigned short *)p.y + ybase_off;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + rsub;
        const int fm = m0 + wm * 64 + row;
        u32x4_t v = *(const u32x4_t *)(wbuf + row * ROWP + ch * 16);
        if (has_res) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                float lo = Tr::to_f32((unsigned short)(v[d] & 0xffffu)) + Tr::to_f32((unsigned short)(rv[it][d] & 0xffffu));
                float hi = Tr::to_f32((unsigned short)(v[d] >> 16)) + Tr::to_f32((unsigned short)(rv[it][d] >> 16));
                if (p.act == 1) {
                    lo = lo > 0.f ? lo : 0.f;
                    hi = hi > 0.f ? hi : 0.f;
                } else if (p.act != 0) {
                    lo = apply_act(lo, p.act);
                    hi = apply_act(hi, p.act);
                }
                v[d] = Tr::pack2(lo, hi);
            }
        }
        if (p.epi_probe == 2) { // ablation hooks (IROCM_CONV_EPI_PROBE): 2 = no global stores
            if (v[0] == 0x12345678u && v[3] == 0x9abcdef0u)
                ((unsigned short *)p.y)[0] = 1;
            continue;
        }
        if (p.epi_probe == 3 && it >= 4) // 3 = half of the rows
            continue;
        if (live && fm < p.f) {
            unsigned short *dst = ybase + (long)fm * p.hw;
            if (p.epi_probe == 4) // 4 = same pattern folded into 1 MiB (cache-resident destination)
                dst = (unsigned short *)p.y + (((dst - (unsigned short *)p.y)) & 0x7ffff);
            if (full) {
                *(u32x4_t *)dst = v;
            } else { // the last run of a plane whose size is not a multiple of 8
                const int nv = p.hw - pix;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (e < nv)
                        dst[e] = (unsigned short)(v[e >> 1] >> ((e & 1) * 16));
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

template <typename Tr, int WM, int WN, int BK, bool ROWTAP>
__global__ __launch_bounds__(256, (BK == 32 ? 3 : 2)) void conv_s1_kernel(ConvS1Args p) {
    constexpr int BM = WM * 64, BN = WN * 64, APITCH = BK + 8;
    constexpr int A_BYTES = BM * APITCH * 2, ROWB = BN * 2, B_BYTES = BK * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int NA = BM * (BK / 8) / 256;       // 16-byte A runs per thread per K-step
    constexpr int CPR = BN / 8;                   // B runs per k-row
    constexpr int KSTEP = 256 / CPR;              // k-rows covered by one pass of the workgroup
    constexpr int NB = BK / KSTEP;                // B runs per thread per K-step
    static_assert(WM * WN == 4 && NA >= 1 && NB >= 1, "4 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w / WN, wn = w % WN;
    const int l15 = lane & 15, g4 = lane >> 4;
    const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = wg % p.tiles_m, tn = wg / p.tiles_m; // filter tiles fastest: neighbours share the B tile
    const int m0 = tm * BM, n0 = tn * BN;

    const unsigned short *Wp = (const unsigned short *)p.w;
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.x), 0, (int)p.x_bytes, 0x00020000);

    // ---- A staging assignment: run ch = t + i*256 -> row ch / (BK/8), k-chunk ch % (BK/8) ------------
    long a_off[NA]; // element offset inside one tap's [F][C] matrix
    int a_lds[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int ch = t + i * 256;
        const int row = ch / (BK / 8), kc = (ch % (BK / 8)) * 8;
        int gm = m0 + row;
        gm = gm < p.f ? gm : p.f - 1; // rows past F re-read the last filter; never stored
        a_off[i] = (long)gm * (ROWTAP ? p.kpad : p.c) + kc;
        a_lds[i] = (row * APITCH + kc) * 2;
    }
    // ---- B staging assignment: one column run per thread, k-rows t / CPR + i * KSTEP ------------------
    const int cchunk = t % CPR, krow0 = t / CPR;
    const int col8 = n0 + cchunk * 8;
    int img, pp;
    fast_divmod(col8, p.hwp, p.hwp_m, img, pp);
    // validity bit sets: rowm bit (8*r + j) = pixel j of the run has input row oh + r - ph inside the image.
    // (Selects, not branches, and multiply-high divisions: this block was a few thousand cycles of every workgroup's prologue.)
    unsigned long rowm = 0, colm = 0;
    {
        const bool run_live = col8 < p.ncols;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int pix = pp + j;
            const bool live = run_live && pix < p.hw;
            int oh, ow;
            fast_divmod(pix < p.hw ? pix : 0, p.wd, p.wd_m, oh, ow);
            for (int r = 0; r < p.r; ++r) // (wave-uniform trip counts)
                rowm |= (unsigned long)((live && (unsigned)(oh * p.sh + r * p.dh - p.ph) < (unsigned)p.in_h) ? 1 : 0) << (8 * r + j);
            for (int s = 0; s < p.s; ++s)
                colm |= (unsigned long)((live && (unsigned)(ow * p.sw + s * p.dw - p.pw) < (unsigned)p.in_w) ? 1 : 0) << (8 * s + j);
        }
    }
    const int b_base = (int)((((long)img * p.c + (ROWTAP ? 0 : krow0)) * p.hw + pp) * 2); // bytes; x_bytes < 2^31
    int b_lds[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int kr = krow0 + i * KSTEP;
        b_lds[i] = kr * ROWB + ((cchunk ^ (f128::mn_f(kr) << 1)) * 16);
    }

    // ROWTAP: k -> (offset, tap selectors), once per workgroup instead of four integer divisions per row and K-step
    int2 *ktab = reinterpret_cast<int2 *>(smem + 2 * STAGE);
    if constexpr (ROWTAP) {
        for (int k = t; k < p.kpad; k += 256) {
            int2 e = {0, 0}; // bit 16 of y = valid
            if (k < p.kdim) {
                const int tp = k / p.c, cc = k - tp * p.c;
                const int r_ = tp / p.s, s_ = tp - r_ * p.s;
                const int dy = r_ * p.dh - p.ph, dx = s_ * p.dw - p.pw;
                const int qy = dy >= 0 ? dy / p.sh : -((p.sh - 1 - dy) / p.sh), py = dy - qy * p.sh;
                const int qx = dx >= 0 ? dx / p.sw : -((p.sw - 1 - dx) / p.sw), px = dx - qx * p.sw;
                e.x = (int)((((long)p.slot[py * p.sw + px] * p.plane_elems + qy * p.wd + qx) + (long)cc * p.hw) * 2);
                e.y = (8 * r_) | ((8 * s_) << 8) | (1 << 16);
            }
            ktab[k] = e;
        }
        __syncthreads();
    }

    s16x8_t a_reg[NA];
    u32x4_t b_reg[NB];
    unsigned bm[4];        // and-masks of the runs held in b_reg (one tap per K-step)
    unsigned m8row[NB];    // ROWTAP: 8-bit validity per row
    int tap = 0, rr = 0, ss = 0, cb = 0; // position of the NEXT K-step to load
    int kl = 0;                          // ROWTAP: index of the next K-step to load
    int tap_shift = 0;                   // bytes
    const long tap_stride = (long)p.f * p.c;
    const int kstep_bytes = KSTEP * p.hw * 2;
    // tap -> (phase plane, shift inside it) in bytes: input row oh*sh + dy = (oh + qy)*sh + py
    auto shift_of = [&](int r_, int s_) {
        const int dy = r_ * p.dh - p.ph, dx = s_ * p.dw - p.pw;
        const int qy = dy >= 0 ? dy / p.sh : -((p.sh - 1 - dy) / p.sh), py = dy - qy * p.sh;
        const int qx = dx >= 0 ? dx / p.sw : -((p.sw - 1 - dx) / p.sw), px = dx - qx * p.sw;
        return (int)(((long)p.slot[py * p.sw + px] * p.plane_elems + qy * p.wd + qx) * 2);
    };
    auto expand = [&](unsigned m8, unsigned (&m)[4]) {
#pragma unroll
        for (int d = 0; d < 4; ++d)
            m[d] = ((m8 >> (2 * d)) & 1u) * 0xffffu | ((m8 >> (2 * d + 1)) & 1u) * 0xffff0000u;
    };
    auto set_tap = [&]() {
        if constexpr (!ROWTAP) {
            expand((unsigned)((rowm >> (8 * rr)) & (colm >> (8 * ss)) & 0xff), bm);
            tap_shift = shift_of(rr, ss);
        }
    };
    set_tap();
    const int ncb = p.c / BK;
    auto fetch_run = [&](int voff, bool any) {
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (any) {
            if ((unsigned)voff <= p.x_bytes - 16) {
                v = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, 0, 0);
            } else {
                // run straddles the start or the end of the tensor: the range check of a 16-byte load works on
                // whole (possibly misaligned) dwords and a negative offset voids all of it — fetch by element
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int eo = voff + 2 * j;
                    const unsigned e = (unsigned)eo < p.x_bytes ? (unsigned)__builtin_amdgcn_raw_buffer_load_b16(xrs, eo, 0, 0) : 0u;
                    v[j >> 1] |= e << ((j & 1) * 16);
                }
            }
        }
        return v;
    };
    // SAFE = false: branch-free — one 16-byte buffer load per run whatever its mask (a dead run may read anything: the
    // descriptor keeps it inside the tensor or returns 0, the mask erases it). Valid when no live run of this
    // workgroup can touch bytes outside the tensor (wg_risky below). Divergent branches around the loads make hipcc
    // drain vmcnt(0) before the LDS reads of the same K-step, i.e. serialise fetch and MFMA (measured: 75 -> 61 us
    // with either half removed, 37 us with both).
    auto load_tile = [&](auto safec) __attribute__((always_inline)) {
        constexpr bool SAFE = decltype(safec)::value;
        if constexpr (ROWTAP) {
            const unsigned short *wsrc = Wp + kl * BK;
#pragma unroll
            for (int i = 0; i < NA; ++i)
                a_reg[i] = *(const s16x8_t *)(wsrc + a_off[i]);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                // per-k table (built once per workgroup): byte offset of (channel, tap) and the tap's row / column
                // selectors (8 * r, 8 * s, bit 16 = k exists)
                const int k = kl * BK + krow0 + i * KSTEP;
                const int2 e = ktab[k];
                const unsigned m8 = (unsigned)((rowm >> (e.y & 0xff)) & (colm >> ((e.y >> 8) & 0xff)) & 0xff) & ((e.y >> 16) ? 0xffu : 0u);
                const int voff = b_base + e.x;
                m8row[i] = m8;
                if constexpr (SAFE) b_reg[i] = fetch_run(voff, m8 != 0);
                else b_reg[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, 0, 0);
            }
        } else {
            const unsigned short *wsrc = Wp + (long)tap * tap_stride + cb * BK;
#pragma unroll
            for (int i = 0; i < NA; ++i)
                a_reg[i] = *(const s16x8_t *)(wsrc + a_off[i]);
            const int voff0 = b_base + cb * BK * p.hw * 2 + tap_shift;
            const bool any = (bm[0] | bm[1] | bm[2] | bm[3]) != 0;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if constexpr (SAFE) b_reg[i] = fetch_run(voff0 + i * kstep_bytes, any);
                else b_reg[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff0 + i * kstep_bytes, 0, 0);
            }
        }
    };
    auto advance = [&]() { // move to the following K-step; masks follow the tap
        if constexpr (ROWTAP) {
            ++kl;
        } else if (++cb == ncb) {
            cb = 0;
            ++tap;
            if (++ss == p.s) {
                ss = 0;
                ++rr;
            }
            set_tap();
        }
    };
    auto store_tile = [&](char *stage, const unsigned (&m)[4]) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
            *(s16x8_t *)(stage + a_lds[i]) = a_reg[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            u32x4_t v = b_reg[i];
            if constexpr (ROWTAP) {
                unsigned mr[4];
                expand(m8row[i], mr);
                v[0] &= mr[0]; v[1] &= mr[1]; v[2] &= mr[2]; v[3] &= mr[3];
            } else {
                v[0] &= m[0]; v[1] &= m[1]; v[2] &= m[2]; v[3] &= m[3];
            }
            *(u32x4_t *)(stage + A_BYTES + b_lds[i]) = v;
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bias_v[4]; // in flight during the whole K loop
    conv_load_bias<Tr>(p, m0, wm, l15, bias_v);

    // per-lane LDS fragment offsets
    const int a_frag = ((wm * 64 + l15) * APITCH + g4 * 8) * 2; // + i*16*APITCH*2 + ks*64
    int b_frag[2];                                              // [hh]: + ks*32*ROWB, XOR per (j) below
    // tr-read: lane p supplies k-row g4*8 + hh*4 + (p >> 2), 4 columns (p & 3) * 4 of the 16-column block
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
        b_frag[hh] = (g4 * 8 + hh * 4 + (l15 >> 2)) * ROWB + (l15 & 1) * 8;
    const int mnf_lane[2] = {f128::mn_f(g4 * 8 + (l15 >> 2)), f128::mn_f(g4 * 8 + 4 + (l15 >> 2))};

    const int nk = ROWTAP ? p.kpad / BK : p.r * p.s * ncb;
    // Can a LIVE run of this workgroup start before the tensor or end after it? Only in the first / last (image, plane
    // set): bound the tap shifts by (R*dh + 1) rows + (S*dw + 1) columns of the plane.
    const int reach = (p.r * p.dh + 1) * p.wd + p.s * p.dw + 9;
    const bool risky_lane = col8 < p.ncols && ((img == 0 && pp < reach) || (img == p.nimg - 1 && pp + reach > p.hw));
    const bool wg_risky = __syncthreads_or(risky_lane) != 0;

    auto compute = [&](const char *cur) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            s16x8_t af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i] = *(const s16x8_t *)(cur + a_frag + i * 16 * APITCH * 2 + ks * 64);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cblk = (wn * 64 + j * 16) >> 3; // 16-byte chunk index of the block's first column
                s16x4_t h[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    // k-row bits 5.. (ks) do not enter mn_f: the lane's swizzle term is K-step invariant
                    const int c16 = (cblk + ((l15 >> 1) & 1)) ^ (mnf_lane[hh] << 1);
                    const char *addr = cur + A_BYTES + ks * 32 * ROWB + b_frag[hh] + c16 * 16;
                    h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t *)(addr));
                }
                bf[j] = s16x8_t{h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = Tr::mfma(bf[j], af[i], acc[i][j]);
        }
    };
    auto sweep = [&](auto safec) __attribute__((always_inline)) {
        unsigned m_cur[4];
        load_tile(safec);
#pragma unroll
        for (int d = 0; d < 4; ++d) m_cur[d] = bm[d];
        advance();
        store_tile(smem, m_cur);
        __syncthreads();
        for (int kt = 0; kt + 1 < nk; ++kt) { // steady state: the next K-step always exists
            const char *cur = smem + (kt & 1) * STAGE;
            char *nxt = smem + ((kt + 1) & 1) * STAGE;
            load_tile(safec); // in flight during the MFMAs below: nothing but straight-line code may sit between the
                              // loads and the LDS reads (any branch there makes hipcc drain vmcnt(0) at the join)
#pragma unroll
            for (int d = 0; d < 4; ++d) m_cur[d] = bm[d];
            compute(cur);
            store_tile(nxt, m_cur);
            advance(); // tap / channel-block bookkeeping (scalar branches, divisions) AFTER the stores have waited
            __syncthreads();
        }
        compute(smem + ((nk - 1) & 1) * STAGE);
    };
    if (wg_risky)
        sweep(std::true_type{});
    else
        sweep(std::false_type{});

    // even planes: dword-aligned runs (and the residual's buffer loads need them); odd planes (7x7): the rows start on
    // 2-byte boundaries, plain global stores take that (unaligned access mode): wide_epilogue == 2, the default — measured on
    // ResNet-50's 7x7 layers (bs128): 512->2048 67 -> 41 us, 2048->512 61 -> 46, 3x3 512 112 -> 97, 1024->2048/2 110 -> 83
    if ((p.wide_epilogue == 1 && (!p.res || p.y_bytes) && (p.hw & 1) == 0) ||
        (p.wide_epilogue == 2 && ((p.hw & 1) == 0 ? (!p.res || p.y_bytes) : !p.res))) { // workgroup-uniform
        __syncthreads(); // the K-loop's stages are dead: reuse them as the waves' staging images
        conv_tile_epilogue_lds<Tr>(p, acc, bias_v, m0, n0, wm, wn, lane, smem + w * kEpiWaveBytes);
        return;
    }
    conv_tile_epilogue<Tr>(p, acc, m0, n0, wm, wn, l15, g4);
}

// ------------------------------------------------------------------------------------------------
// Pointwise (1x1, pad 0) convolutions with few input channels (C <= 256): ResNet's 64->256, 128->512, 256->1024
// expansions and their siblings. With 1-4 K-steps per 128 x 128 tile the per-tile prologue (first-load latency) and the
// epilogue of conv_s1_kernel are not hidden by anything and the X tile is re-fetched for every filter tile (measured:
// 256->1024 @14x14 bs128 81 us against a ~12 us HBM floor). Here ONE workgroup owns a 128-slot column tile, keeps the
// whole [C][128] input tile RESIDENT in LDS (read from HBM exactly once) and walks over all filter tiles: the weight
// tile of the next K-step / next filter tile is in flight during the MFMAs and during the epilogue of the current tile,
// so the pipeline never drains. Same fragment layouts, MFMA roles and epilogue as conv_s1_kernel<2, 2, 64>.
// LDS: 2 x 18 KiB weight stages + C x 256 B (64 KiB at C = 256).
// ------------------------------------------------------------------------------------------------
// Workgroup barrier that orders LDS traffic only. __syncthreads() also drains vmcnt(0), i.e. waits for the output stores
// of the tile just finished -- with one workgroup per CU that put the whole store latency on every filter tile.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <typename Tr, int NKB>
__global__ __launch_bounds__(256, (NKB <= 2 ? 2 : 1)) void conv_pw_kernel(ConvS1Args p) {
    constexpr int WN = 2, BK = 64, BM = 128, BN = 128, APITCH = BK + 8;
    constexpr int A_BYTES = BM * APITCH * 2, ROWB = BN * 2, B_BYTES = BK * ROWB;
    constexpr int NA = BM * (BK / 8) / 256; // 4
    constexpr int CPR = BN / 8;             // 16 runs per k-row
    constexpr int KSTEP = 256 / CPR;        // 16 k-rows per pass
    constexpr int NB = BK / KSTEP;          // 4
    extern __shared__ __attribute__((aligned(16))) char smem[]; // [ A block 0 .. NKB-1 | B block 0 .. NKB-1 ]
    char *const bres = smem + NKB * A_BYTES;

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w / WN, wn = w % WN;
    const int l15 = lane & 15, g4 = lane >> 4;
    // neighbouring column tiles complete each other's cache lines (rows of 2 * hw bytes are rarely line aligned):
    // keep them on one XCD, i.e. behind one L2
    const int n0 = (int)xcd_remap(blockIdx.x, gridDim.x) * BN;
    const unsigned short *Wp = (const unsigned short *)p.w; // [F][C]
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.x), 0, (int)p.x_bytes, 0x00020000);

    int a_row[NA], a_kc[NA], a_lds[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int ch = t + i * 256;
        a_row[i] = ch / (BK / 8);
        a_kc[i] = (ch % (BK / 8)) * 8;
        a_lds[i] = (a_row[i] * APITCH + a_kc[i]) * 2;
    }
    const int cchunk = t % CPR, krow0 = t / CPR;
    int col8 = n0 + cchunk * 8;
    col8 = col8 < p.ncols ? col8 : 0; // slots past the last image: compute on column 0, never stored
    const int img = col8 / p.hwp, pp = col8 - img * p.hwp;
    // runs are dword aligned (hw even, pp % 8 == 0): the descriptor's range check zeroes whole dwords past the tensor,
    // which can only be pad slots of the last row
    const int b_base = (int)((((long)img * p.c + krow0) * p.hw + pp) * 2);
    const int kstep_bytes = KSTEP * p.hw * 2, kblock_bytes = BK * p.hw * 2;
    int b_lds[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int kr = krow0 + i * KSTEP;
        b_lds[i] = kr * ROWB + ((cchunk ^ (f128::mn_f(kr) << 1)) * 16);
    }
    const int a_frag = ((wm * 64 + l15) * APITCH + g4 * 8) * 2;
    int b_frag[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
        b_frag[hh] = (g4 * 8 + hh * 4 + (l15 >> 2)) * ROWB + (l15 & 1) * 8;
    const int mnf_lane[2] = {f128::mn_f(g4 * 8 + (l15 >> 2)), f128::mn_f(g4 * 8 + 4 + (l15 >> 2))};

    // the input tile: all C rows, once
    {
        u32x4_t b_reg[NKB][NB];
#pragma unroll
        for (int kt = 0; kt < NKB; ++kt)
#pragma unroll
            for (int i = 0; i < NB; ++i)
                b_reg[kt][i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, b_base + kt * kblock_bytes + i * kstep_bytes, 0, 0);
#pragma unroll
        for (int kt = 0; kt < NKB; ++kt)
#pragma unroll
            for (int i = 0; i < NB; ++i)
                *(u32x4_t *)(bres + kt * B_BYTES + b_lds[i]) = b_reg[kt][i];
    }
    // weights: the whole [128][C] tile of the NEXT filter tile is in flight during the MFMAs and the epilogue of the
    // current one (one K-step of lookahead, ~0.3 us, cannot cover an L2 round trip; a tile can)
    s16x8_t a_reg[NKB][NA];
    auto load_tile_a = [&](int tm) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            int gm = tm * BM + a_row[i];
            gm = gm < p.f ? gm : p.f - 1; // rows past F re-read the last filter; never stored
            const unsigned short *src = Wp + (long)gm * p.c + a_kc[i];
#pragma unroll
            for (int kt = 0; kt < NKB; ++kt)
                a_reg[kt][i] = *(const s16x8_t *)(src + kt * BK);
        }
    };
    auto store_tile_a = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int kt = 0; kt < NKB; ++kt)
#pragma unroll
            for (int i = 0; i < NA; ++i)
                *(s16x8_t *)(smem + kt * A_BYTES + a_lds[i]) = a_reg[kt][i];
    };

    f32x4 acc[4][4];
    auto compute = [&](const char *astage, const char *bblk) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            s16x8_t af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i] = *(const s16x8_t *)(astage + a_frag + i * 16 * APITCH * 2 + ks * 64);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cblk = (wn * 64 + j * 16) >> 3;
                s16x4_t h[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int c16 = (cblk + ((l15 >> 1) & 1)) ^ (mnf_lane[hh] << 1);
                    const char *addr = bblk + ks * 32 * ROWB + b_frag[hh] + c16 * 16;
                    h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t *)(addr));
                }
                bf[j] = s16x8_t{h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = Tr::mfma(bf[j], af[i], acc[i][j]);
        }
    };

    const bool lds_epi = p.wide_epilogue >= 1 && (!p.res || p.y_bytes); // hw is even here
    char *const epi = NKB >= 2 ? smem : bres + NKB * B_BYTES; // NKB == 1: 18 KiB of weights is too small, own region
    float bias_v[4], bias_n[4];
    load_tile_a(0);
    conv_load_bias<Tr>(p, 0, wm, l15, bias_n);
    store_tile_a();
    __syncthreads();
    for (int tm = 0; tm < p.tiles_m; ++tm) {
        const bool has_next = tm + 1 < p.tiles_m;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            bias_v[i] = bias_n[i];
        if (has_next) {
            load_tile_a(tm + 1);
            conv_load_bias<Tr>(p, (tm + 1) * BM, wm, l15, bias_n);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NKB; ++kt)
            compute(smem + kt * A_BYTES, bres + kt * B_BYTES);
        if (lds_epi) {
            if constexpr (NKB >= 2)
                lds_barrier(); // the weight blocks are dead until store_tile_a: stage the output tile there
            conv_tile_epilogue_lds<Tr>(p, acc, bias_v, tm * BM, n0, wm, wn, lane, epi + w * kEpiWaveBytes);
        } else {
            conv_tile_epilogue<Tr>(p, acc, tm * BM, n0, wm, wn, l15, g4);
        }
        if (has_next) {
            lds_barrier(); // every wave is done reading this tile's weights (and its staging image)
            store_tile_a();
            lds_barrier();
        }
    }
}


// ------------------------------------------------------------------------------------------------
// conv_patch: unit-stride "same" R x S convolutions (the 3x3 layers) with the input patch RESIDENT in LDS.
//
// conv_s1_kernel re-fetches the B tile from L2 for every tap (nine times the input per 3x3 layer, through registers,
// 2-byte-aligned 16-byte loads; rocprofv3 on 3x3 C256 14x14: MFMA busy 17 %). Here, per block of 32 channels, the
// workgroup loads ONCE the input slots its 128 output slots can reach through any tap — [n0 - halo, n0 + 128 + halo),
// halo = ph * W + pw rounded up to 8 — and all R*S taps are computed from that patch:
//   global (NCHW rows, 16-byte runs of 8 slots) -> registers -> RAW image [32 k][slots] in LDS (512-byte rows, the
//   GEMM's XOR swizzle) -> transposed INSIDE LDS by the gfx950 transpose read (2 x ds_read_b64_tr_b16 give a lane the 8
//   consecutive channels of one slot) -> PM image [slot][32 k] (64-byte rows, chunk ^= ((slot >> 2) & 1) << 1:
//   conflict-free for ds_read_b128 at every row offset). In PM a tap is a ROW offset ((r - ph) * W + (s - pw)): the B
//   fragment of v_mfma_f32_16x16x32 for 16 slots is one aligned ds_read_b128 per lane whatever the tap, zero padding is
//   a per-lane validity bit per (slot, tap) applied with v_cndmask (slots are flat pixel indices img * HWp + pix, as in
//   conv_s1: a tap that leaves the image row / plane is masked, so what the neighbouring slots hold never matters).
// Per tap one weight tile [128 f][32 c] (re-packed [RS][F][C]) arrives by LDS-DMA (global_load_lds_dwordx4): two stages of
// THREE taps each (a filter row of a 3x3), fetched one step ahead; one barrier per three taps (48 MFMAs per wave): a first version staged it through registers one step
// ahead like conv_s1_kernel and ran no faster than the tap-shifted kernel — with 16 MFMAs per step (256 cycles) the step
// was bound by the L2 round trip of the weight loads, not by the B traffic the patch removes. The next channel block's
// runs are in flight during the taps of the current one. 2 workgroups per CU (<= 64 KiB of LDS each). Same accumulator
// layout and epilogues (LDS-staged row-wise stores, bias / residual / activation) as conv_s1_kernel<2, 2, *>.
// ------------------------------------------------------------------------------------------------
template <typename Tr, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 8 ? 1 : 2)) void conv_patch_kernel(ConvS1Args p, int halo8, int pslots) {
    // waves of 64 x 64: <2, 2> = 4 waves, 128 f x 128 slots, two workgroups per CU; <2, 4> = 8 waves, 128 f x 256 slots, one per CU —
    // the same waves per CU, but every weight tile streamed from L2 serves twice the slots (the weight stream paces this kernel)
    constexpr int BM = WM * 64, BN = WN * 64, BK = 32;
    constexpr int NW = WM * WN, NTHR = NW * 64;
    constexpr int A_BYTES = BM * BK * 2;                     // weight tile: [BM f][32 c], 64-byte rows
    constexpr int TPS = 3;                                   // taps per step (= per barrier): a filter row of a 3x3
    constexpr int STAGE_BYTES = TPS * A_BYTES;               // a stage = the tiles of TPS consecutive taps
    // stages in flight: two with 4 waves (fetched one step ahead); three with 8 waves — one workgroup per CU has the LDS for it,
    // and a weight step then has TWO steps (>= 2 x 48 MFMAs per wave) to make its L2 round trip
    constexpr int NSTAGE = (WM * WN == 8) ? 3 : 2, AHEAD = NSTAGE - 1;
    constexpr int RAW_CH = WN == 4 ? 64 : 32;                // 16-byte chunks per RAW row (>= patch slots / 8)
    constexpr int RAW_ROWB = RAW_CH * 16, RAW_BYTES = BK * RAW_ROWB; // [32 k][slots], fixed pitch
    constexpr int NR = (BK * (BN + 128) / 8 + NTHR - 1) / NTHR; // patch runs per thread per channel block (patch <= BN + 128 slots)
    constexpr int NPA = BM / 16 / NW;                        // weight DMA pieces (16 rows x 64 B) per wave and tile
    static_assert((NW == 4 || NW == 8) && NPA >= 1, "4 or 8 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[]; // [A stages | RAW | PM]
    char *const raw = smem + NSTAGE * STAGE_BYTES;
    char *const pm = raw + RAW_BYTES;

    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = w / WN, wn = w % WN;
    const int l15 = lane & 15, g4 = lane >> 4;
    const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = wg % p.tiles_m, tn = wg / p.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const unsigned short *Wp = (const unsigned short *)p.w;
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const int ncb = p.c / BK, ntaps = p.r * p.s;
    const long tap_stride = (long)p.f * p.c;
    const unsigned lds0 = (unsigned)(unsigned long)IROCM_LDS_PTR(smem);

    // ---- weight tile by LDS-DMA: BM / 16 pieces of 16 rows x 64 B; wave w issues pieces NPA w ... The image is lane-linear
    // (row = piece * 16 + lane / 4, 16-byte chunk lane % 4), so the chunk swizzle ((row >> 2) & 1) << 1 that makes the
    // fragment reads conflict-free is applied to the global SOURCE chunk. ------------------------------------------
    unsigned a_off[NPA]; // byte offset of this lane's 16 bytes inside one tap's [F][C] matrix (channel block 0)
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int row = (w * NPA + i) * 16 + (lane >> 2);
        const int c_src = (lane & 3) ^ (((row >> 2) & 1) << 1);
        int gm = m0 + row;
        gm = gm < p.f ? gm : p.f - 1; // rows past F re-read the last filter; never stored
        a_off[i] = (unsigned)(((long)gm * p.c + c_src * 8) * 2);
    }
    // DMA cursor: the next step (channel block d_cb, taps TPS d_st ..) to fetch. Past the end it stays on the last one
    // (re-fetched into a dead stage), a missing tap re-fetches the step's first: every wave always issues exactly NPA TPS DMA
    // instructions per step, so the vmcnt counts below are uniform.
    const int nst = (ntaps + TPS - 1) / TPS;
    int d_cb = 0, d_st = 0;
    auto dma_step = [&](int stage) __attribute__((always_inline)) {
        const int t0 = TPS * d_st;
        const unsigned off0 = (unsigned)(((long)t0 * tap_stride + d_cb * BK) * 2); // weights are < 4 GiB
        char *dst = smem + stage * STAGE_BYTES + w * (NPA * 1024);
#pragma unroll
        for (int k = 0; k < TPS; ++k) {
            const char *src = (const char *)Wp + off0 + (t0 + k < ntaps ? (unsigned)(k * tap_stride * 2) : 0u);
#pragma unroll
            for (int i = 0; i < NPA; ++i)
                __builtin_amdgcn_global_load_lds(IROCM_GLB_PTR(src + (unsigned long)a_off[i]),
                                                 IROCM_LDS_PTR(dst + k * A_BYTES + i * 1024), 16, 0, 0);
        }
        if (d_st + 1 < nst) ++d_st;
        else if (d_cb + 1 < ncb) { d_st = 0; ++d_cb; }
    };
    // ---- patch runs: id = t + i*256 -> channel row id / rpr, slot run id % rpr ---------------------------
    const int rpr = pslots / 8, nruns = BK * rpr;
    const unsigned rpr_m = (unsigned)((1ull << 32) / (unsigned)rpr); // (wave-uniform, once; rpr >= 16)
    const int pstart = n0 - halo8; // first slot of the patch (a multiple of 8, may be negative)
    int p_voff[NR], p_lds[NR];
    unsigned p_valid = 0;
    bool risky_lane = false;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int id = t + i * NTHR;
        int kr, rc;
        fast_divmod(id, rpr, rpr_m, kr, rc);
        const int slot0 = pstart + rc * 8;
        int img = 0, pix = 0;
        bool ok = id < nruns && slot0 >= 0 && slot0 < p.ncols;
        if (ok) {
            fast_divmod(slot0, p.hwp, p.hwp_m, img, pix);
            ok = pix < p.hw;
        }
        p_voff[i] = (int)((((long)img * p.c + kr) * p.hw + pix) * 2);
        p_lds[i] = id < nruns ? kr * RAW_ROWB + ((rc ^ (f128::mn_f(kr) << 1)) * 16) : -1;
        if (ok) {
            p_valid |= 1u << i;
            // can the run end past the tensor in the LAST channel block? (only the last plane of an hw % 8 != 0 layer)
            risky_lane = risky_lane || ((unsigned)(p_voff[i] + (ncb - 1) * BK * p.hw * 2) > p.x_bytes - 16);
        }
    }
    const bool wg_risky = __syncthreads_or(risky_lane) != 0;
    // ---- per-lane validity of (slot, tap) and PM row of the lane's slot for the four 16-slot blocks ------
    // Computed AFTER the first loads are issued (sweep): the timeline (s_memtime stamps, round 3) showed 12.5 k cycles between kernel
    // entry and the first load — 15 % of a workgroup's life — most of them in this block when it was a tap-by-tap loop under
    // divergent conditions. Now: one column mask and one select per filter row, no divergence.
    int vm[4];
    const int brow64 = (halo8 + wn * 64 + l15) * 64; // PM byte offset of the lane's slot in block j = 0 (j adds 1024)
    auto compute_vm = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int slot = n0 + wn * 64 + j * 16 + l15;
            const int sl = slot < p.ncols ? slot : 0;
            int img, pix, oh, ow;
            fast_divmod(sl, p.hwp, p.hwp_m, img, pix);
            const int px = pix < p.hw ? pix : 0;
            fast_divmod(px, p.wd, p.wd_m, oh, ow);
            const bool live = slot < p.ncols && pix < p.hw;
            unsigned colmask = 0;
            for (int s_ = 0; s_ < p.s; ++s_) // (wave-uniform trip count)
                colmask |= ((unsigned)(ow + s_ - p.pw) < (unsigned)p.in_w ? 1u : 0u) << s_;
            unsigned m = 0;
            for (int r_ = 0; r_ < p.r; ++r_)
                m |= ((unsigned)(oh + r_ - p.ph) < (unsigned)p.in_h ? colmask : 0u) << (r_ * p.s);
            vm[j] = live ? (int)m : 0;
        }
    };
    // ---- LDS fragment offsets ----------------------------------------------------------------------------
    // weight fragment of filter row wm*64 + i*16 + l15: chunk g4 of a 64-byte row, swizzled like the DMA source
    const unsigned a_frag = lds0 + (unsigned)((wm * 64 + l15) * 64 + ((g4 ^ (((l15 >> 2) & 1) << 1)) * 16)); // + i * 1024
    const unsigned pm0 = lds0 + NSTAGE * STAGE_BYTES + RAW_BYTES;
    const int g4_16 = g4 * 16;
    int t_frag[2]; // transposing pass: lane supplies k-row g4*8 + hh*4 + (l15 >> 2), 4 slots (l15 & 3) * 4 of the block
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
        t_frag[hh] = (g4 * 8 + hh * 4 + (l15 >> 2)) * RAW_ROWB + (l15 & 1) * 8;
    const int mnf_lane[2] = {f128::mn_f(g4 * 8 + (l15 >> 2)), f128::mn_f(g4 * 8 + 4 + (l15 >> 2))};

    u32x4_t p_reg[NR];
    // SAFE = false: one 16-byte buffer load per run (runs that do not exist get an offset past the descriptor: zeros);
    // SAFE = true (a live run of this workgroup may end past the tensor): such a run is fetched by element, because the
    // descriptor's range check works on whole dwords of a possibly 2-byte-aligned run.
    auto load_patch = [&](auto safec, int cb) __attribute__((always_inline)) {
        constexpr bool SAFE = decltype(safec)::value;
        const int cboff = cb * BK * p.hw * 2;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const bool ok = (p_valid >> i) & 1u;
            const int voff = ok ? p_voff[i] + cboff : (int)0xfffffff0u;
            if constexpr (SAFE) {
                u32x4_t v = {0u, 0u, 0u, 0u};
                if (ok) {
                    if ((unsigned)voff <= p.x_bytes - 16) {
                        v = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, 0, 0);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int eo = voff + 2 * j;
                            const unsigned e = (unsigned)eo < p.x_bytes ? (unsigned)__builtin_amdgcn_raw_buffer_load_b16(xrs, eo, 0, 0) : 0u;
                            v[j >> 1] |= e << ((j & 1) * 16);
                        }
                    }
                }
                p_reg[i] = v;
            } else {
                p_reg[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, 0, 0);
            }
        }
    };
    auto store_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NR; ++i)
            if (p_lds[i] >= 0)
                *(u32x4_t *)(raw + p_lds[i]) = p_reg[i];
    };
    // RAW [k][slots] -> PM [slot][k]: wave w transposes the 16-slot blocks w, w + 4, ...
    auto transpose = [&]() __attribute__((always_inline)) {
        for (int sb = w; sb * 16 < pslots; sb += NW) {
            s16x4_t h[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int c16 = (sb * 2 + ((l15 >> 1) & 1)) ^ (mnf_lane[hh] << 1);
                const char *addr = raw + t_frag[hh] + c16 * 16;
                h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t *)(addr));
            }
            const int row = sb * 16 + l15;
            *(s16x8_t *)(pm + row * 64 + ((g4 ^ (((row >> 2) & 1) << 1)) * 16)) =
                s16x8_t{h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bias_v[4];
    conv_load_bias<Tr>(p, m0, wm, l15, bias_v);

    // The fragment reads are inline asm (like gemm256.hip): hipcc knows nothing about what an LDS-DMA in flight writes
    // and would drain vmcnt(0) in front of every LDS read it can see.
    // One tap: 4 weight fragments + 4 slot fragments (blocks j are 16 rows = 1024 bytes apart and 16 rows never change
    // (row >> 2) & 1: one address, four immediate offsets), validity as an all-ones / all-zeros word per (block, tap).
    auto compute = [&](unsigned abase, int tap, int shift64) __attribute__((always_inline)) {
        s16x8_t af[4];
        u32x4_t bv[4];
        af[0] = g256::lds_read_b128<0>(abase);
        af[1] = g256::lds_read_b128<1024>(abase);
        af[2] = g256::lds_read_b128<2048>(abase);
        af[3] = g256::lds_read_b128<3072>(abase);
        const int r64 = brow64 + shift64;
        const unsigned baddr = pm0 + (unsigned)(r64 + (((r64 >> 3) & 32) ^ g4_16));
        bv[0] = __builtin_bit_cast(u32x4_t, g256::lds_read_b128<0>(baddr));
        bv[1] = __builtin_bit_cast(u32x4_t, g256::lds_read_b128<1024>(baddr));
        bv[2] = __builtin_bit_cast(u32x4_t, g256::lds_read_b128<2048>(baddr));
        bv[3] = __builtin_bit_cast(u32x4_t, g256::lds_read_b128<3072>(baddr));
        g256::wait_lgkm0();
        s16x8_t bf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned m = (unsigned)__builtin_amdgcn_sbfe(vm[j], tap, 1); // 0 or 0xffffffff
            u32x4_t v = bv[j];
            v[0] &= m; v[1] &= m; v[2] &= m; v[3] &= m;
            bf[j] = __builtin_bit_cast(s16x8_t, v);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = Tr::mfma(bf[j], af[i], acc[i][j]);
    };

    auto sweep = [&](auto safec) __attribute__((always_inline)) {
        load_patch(safec, 0);
#pragma unroll
        for (int a = 0; a < AHEAD; ++a)
            dma_step(a);
        compute_vm(); // (under the loads' round trip)
        store_patch(); // (the compiler waits for the runs here — and with them for the weight tiles)
        __syncthreads();
        transpose();
        __syncthreads();
        int stage = 0;
        const int row64 = p.wd * 64;
        for (int cb = 0; cb < ncb; ++cb) {
            int sh64 = (-p.ph * p.wd - p.pw) * 64, ss = 0; // tap (0, 0)
            auto next_tap = [&]() {
                sh64 += 64;
                if (++ss == p.s) { ss = 0; sh64 += row64 - p.s * 64; }
            };
            for (int st = 0; st < nst; ++st) {
                // the tiles of the NEXT step, into the stage the previous step read (every wave is past that barrier);
                // one step (TPS taps, >= 1 k cycles) covers their L2 round trip
                dma_step(stage + AHEAD < NSTAGE ? stage + AHEAD : stage + AHEAD - NSTAGE);
                if (st == 0) {
                    // the next channel block's runs: in flight during the remaining taps (the last block re-reads itself)
                    load_patch(safec, cb + 1 < ncb ? cb + 1 : cb);
                }
                const unsigned abase = a_frag + stage * STAGE_BYTES;
#pragma unroll
                for (int k = 0; k < TPS; ++k) {
                    if (k == 0 || TPS * st + k < ntaps) {
                        compute(abase + k * A_BYTES, TPS * st + k, sh64);
                        next_tap();
                    }
                }
                // my pieces of the next step's tiles have landed: every VMEM op — except the NR runs when they were issued
                // after the tiles in this step. (SAFE issues a data-dependent number of loads for the runs: there the
                // plain vmcnt(0) is the only count that is right for every wave.)
                // (loads return in order: "at most N outstanding" with N = what was issued AFTER the next step's tiles — the tiles of
                // the steps beyond it and, in a block's first step, the NR runs)
                if (decltype(safec)::value) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (st == 0) g256p_wait_vm<(AHEAD - 1) * NPA * TPS + NR>();
                else g256p_wait_vm<(AHEAD - 1) * NPA * TPS>();
                g256::barrier();
                stage = stage + 1 < NSTAGE ? stage + 1 : 0;
            }
            if (cb + 1 < ncb) {
                store_patch();
                __syncthreads();
                transpose();
                __syncthreads();
            }
        }
    };
    if (wg_risky)
        sweep(std::true_type{});
    else
        sweep(std::false_type{});

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the tail's dead weight tiles must not land on the staging images
    __syncthreads();
    if ((p.wide_epilogue == 1 && (!p.res || p.y_bytes) && (p.hw & 1) == 0) ||
        (p.wide_epilogue == 2 && ((p.hw & 1) == 0 ? (!p.res || p.y_bytes) : !p.res))) { // workgroup-uniform
        conv_tile_epilogue_lds<Tr>(p, acc, bias_v, m0, n0, wm, wn, lane, smem + w * kEpiWaveBytes);
        return;
    }
    conv_tile_epilogue<Tr>(p, acc, m0, n0, wm, wn, l15, g4);
}

// ------------------------------------------------------------------------------------------------
// conv_resident: unit-stride "same" R x S layers with F <= 64 filters and C <= 64 channels (ResNet's C64 -> F64 3x3 @56x56: three
// layers, 0.3 ms of the 3.4 ms graph, 6 x their floor on the tap-shifted kernel, which re-fetches its B tile from L2 for every
// tap through registers one step ahead and idles out the round trip: 7 % MFMA busy). The whole re-packed weight tensor
// ([R S][64 f][C] = 72 KiB for 3x3 x 64) stays RESIDENT in LDS; a persistent workgroup (one per CU, 4 waves of 64 f x 64 slots)
// walks 256-slot tiles: per tile the input patch of both 32-channel blocks is brought in ONCE (global -> registers, one tile
// ahead, -> RAW -> transposed PM images, exactly as conv_patch), and the R S x C / 32 tap steps then run with NO global memory
// traffic and NO barrier — fragment reads one step ahead of the MFMAs (counted lgkmcnt; with one wave per SIMD nothing else
// would cover the LDS latency). The prologue (weights, address setup, cold instruction fetch) is paid once per CU, not per tile.
// Planes must be a multiple of 8 pixels (no run ever ends past the tensor; other layers keep the tap-shifted kernel).
// ------------------------------------------------------------------------------------------------
template <typename Tr>
__global__ __launch_bounds__(256, 1) void conv_resident_kernel(ConvS1Args p, int halo8, int pslots, int ntiles) {
    constexpr int BN = 256, BK = 32, NCBMAX = 2;
    constexpr int WT_TILE = 64 * BK * 2;                 // one (tap, channel block) weight tile: [64 f][32 c], 64-byte rows
    constexpr int RAW_ROWB = 64 * 16, RAW_BYTES = BK * RAW_ROWB; // [32 k][<= 512 slots]
    constexpr int NR = (BK * (BN + 128) / 8 + 255) / 256; // patch runs per thread per channel block
    extern __shared__ __attribute__((aligned(16))) char smem[]; // [weights | RAW | PM of block 0 | PM of block 1]
    const int ncb = p.c / BK, ntaps = p.r * p.s, nsteps = ntaps * ncb;
    char *const raw = smem + nsteps * WT_TILE;
    char *const pm = raw + RAW_BYTES;
    const int pm_bytes = pslots * 64;

    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6); // = wn: wave w owns slots [64 w, 64 w + 64) of the tile
    const int l15 = lane & 15, g4 = lane >> 4;
    const unsigned lds0 = (unsigned)(unsigned long)IROCM_LDS_PTR(smem);
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.x), 0, (int)p.x_bytes, 0x00020000);

    // ---- weights: nsteps tiles of 4 pieces (16 rows x 64 B) each, by LDS-DMA, once. Tile index = cb * ntaps + tap. The image is
    // lane-linear (row = piece * 16 + lane / 4, chunk lane % 4); the chunk swizzle ((row >> 2) & 1) << 1 that makes the fragment
    // reads conflict-free is applied to the global SOURCE chunk (as in conv_patch). -------------------------------------------
    {
        const unsigned short *Wp = (const unsigned short *)p.w;
        const long tap_stride = (long)p.f * p.c;
        for (int pc = w; pc < nsteps * 4; pc += 4) { // (wave w takes piece w of every tile)
            const int tile = pc >> 2, q = pc & 3;
            const int cb = tile >= ntaps ? 1 : 0, tap = tile - cb * ntaps; // at most two channel blocks
            const int row = q * 16 + (lane >> 2);
            const int c_src = (lane & 3) ^ (((row >> 2) & 1) << 1);
            const int gm = row < p.f ? row : p.f - 1; // rows past F re-read the last filter; never stored
            const char *src = (const char *)(Wp + tap * tap_stride + (long)gm * p.c + cb * BK + c_src * 8);
            __builtin_amdgcn_global_load_lds(IROCM_GLB_PTR(src), IROCM_LDS_PTR(smem + tile * WT_TILE + q * 1024), 16, 0, 0);
        }
    }
    // ---- per-lane constants ------------------------------------------------------------------------------
    const int rpr = pslots / 8, nruns = BK * rpr;
    const unsigned rpr_m = (unsigned)((1ull << 32) / (unsigned)rpr);
    int run_kr[NR], run_rc[NR], p_lds[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int id = t + i * 256;
        fast_divmod(id, rpr, rpr_m, run_kr[i], run_rc[i]);
        p_lds[i] = id < nruns ? run_kr[i] * RAW_ROWB + ((run_rc[i] ^ (f128::mn_f(run_kr[i]) << 1)) * 16) : -1;
    }
    const unsigned a_frag = lds0 + (unsigned)(l15 * 64 + ((g4 ^ (((l15 >> 2) & 1) << 1)) * 16)); // + tile * WT_TILE + i * 1024
    const unsigned pm0 = lds0 + (unsigned)(nsteps * WT_TILE + RAW_BYTES);
    const int g4_16 = g4 * 16;
    const int brow64 = (halo8 + w * 64 + l15) * 64;
    int t_frag[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
        t_frag[hh] = (g4 * 8 + hh * 4 + (l15 >> 2)) * RAW_ROWB + (l15 & 1) * 8;
    const int mnf_lane[2] = {f128::mn_f(g4 * 8 + (l15 >> 2)), f128::mn_f(g4 * 8 + 4 + (l15 >> 2))};
    float bias_v[4];
    conv_load_bias<Tr>(p, 0, 0, l15, bias_v);

    u32x4_t p_reg[NCBMAX][NR];
    // the patch of tile `tile`: slots [n0 - halo8, n0 + 256 + halo8), both channel blocks; runs that do not exist read zeros.
    // Two phases: the run offsets (patch_addr), then one buffer load per (run, block) (patch_issue) — the tile loop spreads the
    // issues over the tap steps of the previous tile (with one wave per SIMD a burst of twelve loads is twelve issue slots the MFMA
    // pipe idles through).
    int run_voff[NR];
    auto patch_addr = [&](int tile) __attribute__((always_inline)) {
        const int pstart = tile * BN - halo8;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int slot0 = pstart + run_rc[i] * 8;
            const bool ok = p_lds[i] >= 0 && slot0 >= 0 && slot0 < p.ncols;
            int img, pix;
            fast_divmod(ok ? slot0 : 0, p.hwp, p.hwp_m, img, pix); // hwp == hw here (planes are multiples of 8)
            run_voff[i] = ok ? ((img * p.c + run_kr[i]) * p.hw + pix) * 2 : (int)0xfffffff0u; // (x_bytes < 2^31: 32-bit is enough)
        }
    };
    auto patch_issue = [&](auto ic, auto cbc) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value, cb = decltype(cbc)::value;
        if (cb < ncb) {
            const int voff = run_voff[i];
            p_reg[cb][i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff == (int)0xfffffff0u ? voff : voff + cb * BK * p.hw * 2, 0, 0);
        }
    };
    auto load_patch = [&](int tile) __attribute__((always_inline)) { // everything at once (the first tile)
        patch_addr(tile);
        g256::sfor<NR>([&](auto ic) {
            patch_issue(ic, std::integral_constant<int, 0>{});
            patch_issue(ic, std::integral_constant<int, 1>{});
        });
    };
    auto store_patch = [&](int cb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NR; ++i)
            if (p_lds[i] >= 0)
                *(u32x4_t *)(raw + p_lds[i]) = p_reg[cb][i];
    };
    auto transpose = [&](char *pmc) __attribute__((always_inline)) { // RAW [k][slots] -> PM [slot][k], 16-slot blocks w, w + 4, ...
        for (int sb = w; sb * 16 < pslots; sb += 4) {
            s16x4_t h[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int c16 = (sb * 2 + ((l15 >> 1) & 1)) ^ (mnf_lane[hh] << 1);
                const char *addr = raw + t_frag[hh] + c16 * 16;
                h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t *)(addr));
            }
            const int row = sb * 16 + l15;
            *(s16x8_t *)(pmc + row * 64 + ((g4 ^ (((row >> 2) & 1) << 1)) * 16)) =
                s16x8_t{h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
        }
    };
    int vm[4];
    auto compute_vm = [&](int n0) __attribute__((always_inline)) { // 3 x 3, pad 1 (the launcher admits nothing else): nine bits per slot
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int slot = n0 + w * 64 + j * 16 + l15;
            int img, pix, oh, ow;
            fast_divmod(slot < p.ncols ? slot : 0, p.hwp, p.hwp_m, img, pix);
            fast_divmod(pix, p.wd, p.wd_m, oh, ow);
            const unsigned colmask = (ow > 0 ? 1u : 0u) | 2u | (ow + 1 < p.in_w ? 4u : 0u);
            const unsigned m = (oh > 0 ? colmask : 0u) | (colmask << 3) | (oh + 1 < p.in_h ? colmask << 6 : 0u);
            vm[j] = slot < p.ncols ? (int)m : 0;
        }
    };
    struct TapFrags {
        s16x8_t af[4];
        u32x4_t bv[4];
    };
    // One step = 16 MFMAs on the fragments of (tap, block) `step`, with the eight fragment reads of step + 1 issued BETWEEN them
    // (one wave per SIMD: an LDS or VALU instruction issued in front of the MFMA block delays it by its issue slots; placed between two
    // MFMAs it rides in the 16 cycles the matrix pipe needs anyway). Reads are opaque asm, the order is pinned with scheduling
    // fences, and the wait at the top of a step is a plain lgkmcnt(0): only that step's own reads are outstanding then.
    auto frag_addrs = [&](int step, int shift64, unsigned &abase, unsigned &baddr) __attribute__((always_inline)) {
        abase = a_frag + (unsigned)(step * WT_TILE);
        const int r64 = brow64 + shift64;
        baddr = pm0 + (unsigned)((step >= ntaps ? pm_bytes : 0) + r64 + (((r64 >> 3) & 32) ^ g4_16));
    };
    auto read_all = [&](unsigned abase, unsigned baddr, TapFrags &f) __attribute__((always_inline)) {
        f.af[0] = g256::lds_read_b128<0>(abase);
        f.af[1] = g256::lds_read_b128<1024>(abase);
        f.af[2] = g256::lds_read_b128<2048>(abase);
        f.af[3] = g256::lds_read_b128<3072>(abase);
        f.bv[0] = __builtin_bit_cast(u32x4_t, g256::lds_read_b128<0>(baddr));
        f.bv[1] = __builtin_bit_cast(u32x4_t, g256::lds_read_b128<1024>(baddr));
        f.bv[2] = __builtin_bit_cast(u32x4_t, g256::lds_read_b128<2048>(baddr));
        f.bv[3] = __builtin_bit_cast(u32x4_t, g256::lds_read_b128<3072>(baddr));
    };
    f32x4 acc[4][4];
    // MFMAs of `cur` (masked with tap `tap0`), reads into `nxt` interleaved when HAVE_NEXT
    auto step_fn = [&](auto have_next, const TapFrags &cur, int tap0, TapFrags &nxt, unsigned abase, unsigned baddr) __attribute__((always_inline)) {
        constexpr bool HN = decltype(have_next)::value;
        s16x8_t bf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned m = (unsigned)__builtin_amdgcn_sbfe(vm[j], tap0, 1); // 0 or 0xffffffff
            u32x4_t v = cur.bv[j];
            v[0] &= m; v[1] &= m; v[2] &= m; v[3] &= m;
            bf[j] = __builtin_bit_cast(s16x8_t, v);
        }
        g256::fence_sched();
        g256::sfor<16>([&](auto kc) {
            constexpr int k = decltype(kc)::value, i = k >> 2, j = k & 3;
            acc[i][j] = Tr::mfma(bf[j], cur.af[i], acc[i][j]);
            if constexpr (HN && k < 8) {
                g256::fence_sched();
                if constexpr (k == 0) nxt.af[0] = g256::lds_read_b128<0>(abase);
                if constexpr (k == 1) nxt.af[1] = g256::lds_read_b128<1024>(abase);
                if constexpr (k == 2) nxt.af[2] = g256::lds_read_b128<2048>(abase);
                if constexpr (k == 3) nxt.af[3] = g256::lds_read_b128<3072>(abase);
                if constexpr (k == 4) nxt.bv[0] = __builtin_bit_cast(u32x4_t, g256::lds_read_b128<0>(baddr));
                if constexpr (k == 5) nxt.bv[1] = __builtin_bit_cast(u32x4_t, g256::lds_read_b128<1024>(baddr));
                if constexpr (k == 6) nxt.bv[2] = __builtin_bit_cast(u32x4_t, g256::lds_read_b128<2048>(baddr));
                if constexpr (k == 7) nxt.bv[3] = __builtin_bit_cast(u32x4_t, g256::lds_read_b128<3072>(baddr));
                g256::fence_sched();
            }
        });
    };

    int tile = blockIdx.x;
    if (tile < ntiles)
        load_patch(tile);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // weights (and the first patch) have landed
    __syncthreads();
    const int row64 = p.wd * 64;
    for (; tile < ntiles; tile += gridDim.x) {
        const int n0 = tile * BN;
        // ---- patch -> PM images (the registers were filled one tile ago) ----------------------------------
        for (int cb = 0; cb < ncb; ++cb) {
            if (cb == 0) store_patch(0);
            else store_patch(1);
            __syncthreads();
            transpose(pm + cb * pm_bytes);
            __syncthreads();
        }
        const bool more = tile + (int)gridDim.x < ntiles;
        if (more)
            patch_addr(tile + gridDim.x);
        compute_vm(n0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // ---- the tap steps: no global memory, no barrier ---------------------------------------------------
        int sh64 = (-p.ph * p.wd - p.pw) * 64, ss = 0, tap = 0;
        auto next_tap = [&]() {
            sh64 += 64;
            if (++ss == p.s) { ss = 0; sh64 += row64 - p.s * 64; }
            if (++tap == ntaps) { tap = 0; ss = 0; sh64 = (-p.ph * p.wd - p.pw) * 64; }
        };
        // Fully unrolled, branch-free step sequence per supported step count (3 x 3 taps x one or two channel blocks): with a
        // conditional second half in a rolled loop hipcc kept the accumulators in AGPRs on one path and VGPRs on the other and
        // moved all 64 of them across EVERY step (64 v_accvgpr_write + 64 v_accvgpr_read per 16 MFMAs: 970 cycles per step).
        TapFrags fr[2];
        auto run_steps = [&](auto nc) __attribute__((always_inline)) {
            constexpr int N = decltype(nc)::value;
            {
                unsigned ab, bb;
                frag_addrs(0, sh64, ab, bb);
                read_all(ab, bb, fr[0]);
            }
            g256::sfor<N>([&](auto sc) {
                constexpr int step = decltype(sc)::value;
                const int tap0 = tap;
                next_tap();
                unsigned ab = 0, bb = 0;
                if constexpr (step + 1 < N)
                    frag_addrs(step + 1, sh64, ab, bb);
                g256::wait_lgkm0(); // this step's fragments (issued during the previous step's MFMAs)
                step_fn(std::integral_constant<bool, (step + 1 < N)>{}, fr[step & 1], tap0, fr[(step + 1) & 1], ab, bb);
                // the next tile's patch: one (run, block) load behind each of the first 2 NR steps (N >= NR for one block)
                if constexpr (step < 2 * NR) {
                    if (more) {
                        if constexpr (N >= 2 * NR) patch_issue(std::integral_constant<int, step / 2>{}, std::integral_constant<int, step % 2>{});
                        else if constexpr (step < NR) patch_issue(std::integral_constant<int, step>{}, std::integral_constant<int, 0>{});
                    }
                }
                g256::fence_sched();
            });
        };
        if (nsteps == 18) run_steps(std::integral_constant<int, 18>{});
        else run_steps(std::integral_constant<int, 9>{}); // (the launcher admits 9 and 18 only)
        // ---- epilogue: staged through RAW + PM (dead now; 36 KiB for the four waves) ---------------------------
        __syncthreads();
        conv_tile_epilogue_lds<Tr>(p, acc, bias_v, 0, n0, 0, w, lane, raw + w * kEpiWaveBytes);
        __syncthreads();
    }
}

template <typename Tr> static int launch_resident(infiniRocmRuntime_t rt, ConvS1Args &p, int halo8) {
    const int pslots = 256 + 2 * halo8;
    const int ncb = p.c / 32, nsteps = p.r * p.s * ncb;
    const int lds = nsteps * 4096 + 32 * 1024 + ncb * pslots * 64;
    const int ntiles = (int)ceil_div(p.ncols, 256);
    p.tiles_m = 1;
    p.tiles_n = ntiles;
    auto kern = conv_resident_kernel<Tr>;
    IROCM_LDS_ATTR(kern, 160 * 1024, rt);
    const int grid = ntiles < rt->num_cu ? ntiles : rt->num_cu;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, rt->stream, p, halo8, pslots, ntiles);
    IROCM_LAUNCH_CHECK("conv_resident");
    rt->last_conv_route = "resident";
    return INFINI_ROCM_OK;
}

template <typename Tr, int WM, int WN> static int launch_patch(infiniRocmRuntime_t rt, ConvS1Args &p, int halo8, int pslots) {
    constexpr int BM = WM * 64, BN = WN * 64;
    p.tiles_m = (int)ceil_div(p.f, BM);
    p.tiles_n = (int)ceil_div(p.ncols, BN);
    const long blocks = (long)p.tiles_m * p.tiles_n;
    IROCM_CHECK_ARG(blocks < (1l << 31), "conv2d: too many tiles");
    constexpr int fixed = (WM * WN == 8 ? 3 : 2) * 3 * (BM * 32 * 2) + 32 * (WN == 4 ? 1024 : 512); // weight stages + RAW
    const int lds = fixed + pslots * 64;                                         // + PM (<= 80 KiB: two per CU)
    auto kern = conv_patch_kernel<Tr, WM, WN>;
    IROCM_LDS_ATTR(kern, fixed + (BN + 128) * 64, rt);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WM * WN * 64), lds, rt->stream, p, halo8, pslots);
    IROCM_LAUNCH_CHECK("conv_patch");
    return INFINI_ROCM_OK;
}

template <typename Tr, int NKB> static int launch_pw_n(infiniRocmRuntime_t rt, ConvS1Args &p) {
    constexpr int LDS = NKB * (128 * 72 * 2 + 64 * 256) + (NKB == 1 ? 4 * kEpiWaveBytes : 0);
    auto kern = conv_pw_kernel<Tr, NKB>;
    IROCM_LDS_ATTR(kern, LDS, rt);
    hipLaunchKernelGGL(kern, dim3((unsigned)p.tiles_n), dim3(256), LDS, rt->stream, p);
    IROCM_LAUNCH_CHECK("conv_pw");
    return INFINI_ROCM_OK;
}
template <typename Tr> static int launch_pw(infiniRocmRuntime_t rt, ConvS1Args &p) {
    p.tiles_m = (int)ceil_div(p.f, 128);
    p.tiles_n = (int)ceil_div(p.ncols, 128);
    switch (p.c / 64) {
    case 1: return launch_pw_n<Tr, 1>(rt, p);
    case 2: return launch_pw_n<Tr, 2>(rt, p);
    case 3: return launch_pw_n<Tr, 3>(rt, p);
    default: return launch_pw_n<Tr, 4>(rt, p);
    }
}

template <typename Tr, int WM, int WN, int BK, bool ROWTAP = false>
static int launch_s1(infiniRocmRuntime_t rt, ConvS1Args &p) {
    constexpr int BM = WM * 64, BN = WN * 64;
    constexpr int LDS = 2 * (BM * (BK + 8) * 2 + BK * BN * 2);
    p.tiles_m = (int)ceil_div(p.f, BM);
    p.tiles_n = (int)ceil_div(p.ncols, BN);
    const long blocks = (long)p.tiles_m * p.tiles_n;
    IROCM_CHECK_ARG(blocks < (1l << 31), "conv2d: too many tiles");
    auto kern = conv_s1_kernel<Tr, WM, WN, BK, ROWTAP>;
    IROCM_LDS_ATTR(kern, LDS + (ROWTAP ? 16384 : 0), rt);
    const int lds = LDS + (ROWTAP ? p.kpad * 8 : 0); // + the per-k table
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, rt->stream, p);
    IROCM_LAUNCH_CHECK("conv_s1");
    return INFINI_ROCM_OK;
}

// Returns -1 when the shape is not a conv_s1 shape (caller falls through to the generic kernel).
int launch_conv_s1(infiniRocmRuntime_t rt, int dtype, const void *x, const void *w, const void *bias, const void *res,
                   void *y, int n, int c, int h, int wd, int f, int r, int s, int ph, int pw, int sh, int sw, int dh, int dw,
                   int oh, int ow, int act) {
    if (r > 7 || s > 7 || ((uintptr_t)w & 15) != 0 || ((uintptr_t)x & 3) != 0)
        return -1;
    const bool rowtap = c % 32 != 0;
    if (rowtap && (((long)c * r * s + 31) & ~31l) > 2048)
        return -1; // the per-k LDS table holds 2048 entries
    if (sh * sw > 16 || oh != (h + sh - 1) / sh || ow != (wd + sw - 1) / sw)
        return -1;
    ConvS1Args p;
    p.x = x; p.w = w; p.bias = bias; p.res = res; p.y = y;
    p.nimg = n; p.c = c; p.f = f; p.r = r; p.s = s; p.ph = ph; p.pw = pw;
    p.sh = sh; p.sw = sw; p.dh = dh; p.dw = dw;
    p.in_h = h; p.in_w = wd; p.h = oh; p.wd = ow;
    p.hw = oh * ow;
    p.hwp = (p.hw + 7) & ~7;
    p.hwp_m = divmod_magic(p.hwp);
    p.wd_m = divmod_magic(ow);
    if ((long)n * p.hwp >= (1l << 31))
        return -1;
    p.ncols = n * p.hwp;
    p.act = act;
    static const int wide = getenv("IROCM_CONV_WIDE") ? atoi(getenv("IROCM_CONV_WIDE")) : 2; // 0 off, 1 even planes only, 2 all
    p.wide_epilogue = wide;
    static const int epi_probe = getenv("IROCM_CONV_EPI_PROBE") ? atoi(getenv("IROCM_CONV_EPI_PROBE")) : 0;
    p.epi_probe = epi_probe;
    p.plane_elems = (long)n * c * p.hw;
    {
        const long yb = (long)n * f * p.hw * 2;
        p.y_bytes = (yb < (1l << 31) - 64 && (((uintptr_t)res) & 3) == 0) ? (unsigned)yb : 0u; // else: the direct epilogue
    }
    // phases read by some tap
    PhaseSplitArgs ps;
    ps.nslots = 0;
    for (int i = 0; i < 16; ++i)
        p.slot[i] = -1;
    for (int rr = 0; rr < r; ++rr)
        for (int ss = 0; ss < s; ++ss) {
            const int py = ((rr * dh - ph) % sh + sh) % sh, px = ((ss * dw - pw) % sw + sw) % sw;
            if (p.slot[py * sw + px] < 0) {
                ps.py[ps.nslots] = (signed char)py;
                ps.px[ps.nslots] = (signed char)px;
                p.slot[py * sw + px] = (signed char)ps.nslots++;
            }
        }
    // Round 5: 3 x 3 / pad 1 layers of stride 1 or 2 with >= 256 filters as ONE GEMM with K = 9 C on the persistent 256-row kernels
    // (gemm256p_kernel.h, CONV = 3: TAP mode; gemm256p_conv3.hip). Conv variant 7 forces it for every eligible shape (tests, tune(),
    // tools/conv_bench.py). Default routing by measurement (batch 128, f16, us; tools/conv_bench.py on three boxes):
    //   strided layers (the tap-shifted kernel on phase planes was their only kernel): C256 28 x 28 / 2 -> 74-77 vs 87-92,
    //     C512 14 x 14 / 2 -> 78-82 vs 115-120 (split-K x 4): taken whenever F >= 256;
    //   unit-stride layers compete with the patch kernels: C512 7 x 7 (56 tiles, split-K x 4) 58-61 vs 69-73: taken; C256 14 x 14 (100
    //     tiles, split-K x 2) 51-57 vs 49.5: not taken — i.e. only where the tiles are so few that the split is by four.
    const bool tap_shape = r == 3 && s == 3 && ph == 1 && pw == 1 && dh == 1 && dw == 1 && ((sh == 1 && sw == 1) || (sh == 2 && sw == 2)) &&
                           c % 64 == 0 && !res && (act == 0 || act == 1) && (long)oh * ow >= 8;
    static const int tap_on = getenv("IROCM_CONV_TAP") ? atoi(getenv("IROCM_CONV_TAP")) : 1; // A/B hook: 0 = off
    bool tap_want = tap_shape && rt->conv_variant == 7;
    // (strided layers from 128 filters on: half of the 256-row tile is empty there, and the tap GEMM still beats the phase-plane
    // tap-shifted kernel — C128 -> 128 @56^2 / 2 at batch 128: 98.2 vs 104.7 us; unit-stride layers need the 256 filters)
    if (tap_shape && rt->conv_variant < 0 && tap_on && (f >= 256 || (sh == 2 && f >= 128))) {
        const long tiles256 = ceil_div(f, 256) * ceil_div((long)n * (((long)oh * ow + 7) / 8 * 8), 256);
        if (sh == 2)
            tap_want = tiles256 * 4 >= rt->num_cu / 2;                       // enough work for the persistent kernels at all
        else
            tap_want = conv_tap_split(rt, n, (long)oh * ow, c, f, nullptr) >= 4; // few tiles, long K
    }
    if (tap_want && sh == 2) { // the tap mode addresses the phase planes in the fixed order py * 2 + px
        ps.nslots = 4;
        for (int i = 0; i < 4; ++i) {
            p.slot[i] = (signed char)i;
            ps.py[i] = (signed char)(i >> 1);
            ps.px[i] = (signed char)(i & 1);
        }
    }
    const bool split = sh * sw > 1;
    const long x_bytes = p.plane_elems * 2 * (split ? ps.nslots : 1);
    if (x_bytes >= (1l << 31) - 64 || x_bytes < 64) // 32-bit buffer offsets; `x_bytes - 16` must not wrap
        return -1;
    p.x_bytes = (unsigned)x_bytes;
    // Re-packed weights: FCRS -> [RS][F][C] (or [F][Kpad] for ROWTAP). When the caller declared the weights constant
    // (infini_rocm_conv2d_set_const_weights: the plugin does for graph weights / inputs no operator writes) the packed
    // image is built ONCE, kept in a runtime-owned buffer keyed by (pointer, F, C, RS, layout) and dropped when anything
    // is copied over the source range (runtime.hip) — the reference's cuDNN path has no per-call weight transform either
    // (src/kernels/cuda/conv.cc:143-168). Otherwise it is rebuilt per call in the workspace: [ weights | phase planes ].
    p.kdim = c * r * s;
    p.kpad = (p.kdim + 31) & ~31;
    const size_t w_bytes = rowtap ? (((size_t)f * p.kpad * 2 + 255) & ~(size_t)255)
                                  : (r * s > 1 ? (((size_t)f * c * r * s * 2 + 255) & ~(size_t)255) : 0);
    const bool cached = w_bytes && rt->conv_const_weights;
    const void *packed = nullptr;
    hipStream_t pack_stream = rt->stream; // (may be the legacy default stream, i.e. a null handle: never test it)
    bool need_pack = w_bytes != 0;
    if (cached) {
        packed = wcache_lookup(rt, w, f, c, r * s, rowtap ? 1 : 0);
        if (!packed) {
            void *buf = nullptr;
            int st = wcache_insert(rt, w, (size_t)f * c * r * s * 2, f, c, r * s, rowtap ? 1 : 0, w_bytes, &buf, &pack_stream);
            if (st != INFINI_ROCM_OK)
                return st;
            packed = buf;
        } else {
            need_pack = false; // hit: nothing to launch
        }
    }
    const size_t ws_w = cached ? 0 : w_bytes;
    // the tap GEMM's split-K exchange slab (fp32 partial row blocks) sits behind the weights / phase planes
    size_t tap_slab_bytes = 0;
    const int tap_split = tap_want ? conv_tap_split(rt, n, (long)oh * ow, c, f, &tap_slab_bytes) : 1;
    // (+256: the pixel-slot GEMM a strided pointwise layer continues with reads up to 14 bytes past a ragged last plane)
    const size_t ws_planes = ws_w + (split ? (size_t)x_bytes + 256 : 0);
    const size_t ws_slab_off = (ws_planes + 255) & ~(size_t)255;
    const size_t ws_bytes = tap_slab_bytes ? ws_slab_off + tap_slab_bytes : ws_planes;
    char *ws = nullptr;
    if (ws_bytes) {
        int st = infini_rocm_workspace(rt, ws_bytes, (void **)&ws);
        if (st != INFINI_ROCM_OK)
            return st;
    }
    unsigned short *wdst = cached ? (unsigned short *)const_cast<void *>(packed) : (unsigned short *)ws;
    if (need_pack) {
        // a cache entry published by wcache_insert above must not outlive a failed pack: the next conv with the same key
        // would hit it and read an uninitialised image
        auto fail_pack = [&](const char *what, hipError_t e) {
            if (cached)
                wcache_forget(rt, packed);
            IROCM_FAIL(INFINI_ROCM_HIP_ERROR, "launch of %s failed: %s", what, hipGetErrorString(e));
        };
        if (rowtap) { // FCRS -> [F][Kpad], k = tap * C + c
            long g = ceil_div((long)f * p.kpad, 256);
            if (g > 4096) g = 4096;
            hipLaunchKernelGGL(conv_repack_w_flat, dim3((unsigned)g), dim3(256), 0, pack_stream, (const unsigned short *)w, wdst, f,
                               c, r * s, p.kpad);
            if (hipError_t e = hipGetLastError(); e != hipSuccess)
                return fail_pack("conv_repack_w_flat", e);
        } else { // FCRS -> [RS][F][C]
            long g = ceil_div((long)f * c * r * s, 256);
            if (g > 4096) g = 4096;
            hipLaunchKernelGGL(conv_repack_w, dim3((unsigned)g), dim3(256), 0, pack_stream, (const unsigned short *)w, wdst, f, c,
                               r * s);
            if (hipError_t e = hipGetLastError(); e != hipSuccess)
                return fail_pack("conv_repack_w", e);
        }
        if (cached) {
            int st = wcache_commit(rt, pack_stream);
            if (st != INFINI_ROCM_OK) {
                wcache_forget(rt, packed);
                return st;
            }
        }
    }
    if (w_bytes)
        p.w = wdst;
    const size_t w_off = ws_w;
    if (split) {
        ps.x = (const unsigned short *)x;
        ps.o = (unsigned short *)(ws + w_off);
        ps.planes = (long)n * c;
        ps.in_h = h; ps.in_w = wd; ps.oh = oh; ps.ow = ow; ps.sh = sh; ps.sw = sw;
        const long work2 = (long)n * c * 2 * oh * ((ow + 1) / 2);
        // vector kernels: 8 columns per thread when the row length allows (any phase set), else the quad kernel when all
        // four phases are wanted (3x3/2, 7x7/2); a 1x1/2 on odd-sized rows reads one phase and is faster element-wise
        const bool v8ok = wd % 8 == 0 && (((uintptr_t)x) & 15) == 0 && (w_off % 8 == 0) && (p.plane_elems % 4 == 0);
        // one phase, (0, 0), of an even-width input (the 1 x 1 / 2 layers): the subsampling kernel
        const int vec = wd % 8 == 0 ? 8 : (wd % 4 == 0 ? 4 : (wd % 2 == 0 ? 2 : 0));
        if (sh == 2 && sw == 2 && ps.nslots == 1 && ps.py[0] == 0 && ps.px[0] == 0 && vec && (((uintptr_t)x) & 3) == 0 && (w_off % 4 == 0) &&
            (long)n * c * oh * (wd / vec) < (1l << 31)) {
            Subsample2Args sa;
            sa.x = ps.x; sa.o = ps.o; sa.planes = n * c; sa.in_h = h; sa.in_w = wd; sa.oh = oh; sa.ow = ow;
            sa.groups_m = divmod_magic(wd / vec);
            sa.oh_m = divmod_magic(oh);
            long g = ceil_div((long)n * c * oh * (wd / vec), 256);
            if (g > (long)rt->num_cu * 32) g = (long)rt->num_cu * 32;
            if (vec == 8) hipLaunchKernelGGL(conv_subsample2_kernel<8>, dim3((unsigned)g), dim3(256), 0, rt->stream, sa);
            else if (vec == 4) hipLaunchKernelGGL(conv_subsample2_kernel<4>, dim3((unsigned)g), dim3(256), 0, rt->stream, sa);
            else hipLaunchKernelGGL(conv_subsample2_kernel<2>, dim3((unsigned)g), dim3(256), 0, rt->stream, sa);
        } else if (sh == 2 && sw == 2 && (ps.nslots == 4 || v8ok) && work2 + (long)rt->num_cu * 32 * 256 < (1l << 31)) {
            PhaseSplit2Args a2;
            a2.x = ps.x; a2.o = ps.o; a2.planes = n * c; a2.in_h = h; a2.in_w = wd; a2.oh = oh; a2.ow = ow;
            a2.plane_elems = p.plane_elems;
            for (int i = 0; i < 4; ++i)
                a2.slot[i] = p.slot[i];
            const bool v8 = wd % 8 == 0 && (((uintptr_t)a2.x) & 15) == 0 && (((uintptr_t)a2.o) & 7) == 0 &&
                            (p.plane_elems % 4 == 0);
            long g = ceil_div(v8 ? work2 / 2 : work2, 256);
            if (g > (long)rt->num_cu * 32) g = (long)rt->num_cu * 32;
            const int vn = v8 ? 0 : (wd % 4 == 0 && p.plane_elems % 2 == 0 ? 4 : (wd % 2 == 0 ? 2 : 0));
            if (v8) {
                hipLaunchKernelGGL(conv_phase_split_2x2_v8, dim3((unsigned)g), dim3(256), 0, rt->stream, a2);
            } else if (vn && (((uintptr_t)a2.x) & 3) == 0 && (((uintptr_t)a2.o) & 3) == 0 && (long)n * c * 2 * oh * (wd / vn) < (1l << 31)) {
                PhaseSplit2vArgs av;
                av.a = a2;
                av.groups_m = divmod_magic(wd / vn);
                av.rows_m = divmod_magic(2 * oh);
                long gv = ceil_div((long)n * c * 2 * oh * (wd / vn), 256);
                if (gv > (long)rt->num_cu * 32) gv = (long)rt->num_cu * 32;
                if (vn == 4) hipLaunchKernelGGL(conv_phase_split_2x2_vn<4>, dim3((unsigned)gv), dim3(256), 0, rt->stream, av);
                else hipLaunchKernelGGL(conv_phase_split_2x2_vn<2>, dim3((unsigned)gv), dim3(256), 0, rt->stream, av);
            } else {
                hipLaunchKernelGGL(conv_phase_split_2x2, dim3((unsigned)g), dim3(256), 0, rt->stream, a2);
            }
        } else {
            long g = ceil_div(p.plane_elems * ps.nslots, 256);
            if (g > (long)rt->num_cu * 32) g = (long)rt->num_cu * 32;
            hipLaunchKernelGGL(conv_phase_split, dim3((unsigned)g), dim3(256), 0, rt->stream, ps);
        }
        IROCM_LAUNCH_CHECK("conv_phase_split");
        p.x = ps.o;
        // a strided 1 x 1 layer (ResNet's down-sampling branches) reads ONE phase: the plane set just written is a dense
        // [n][c][oh][ow] activation, i.e. a unit-stride pointwise layer — one GEMM over pixel slots on the persistent kernels
        // when it has the filters to fill their 256-row tiles (conv.hip has the rule and the numbers)
        if (r == 1 && s == 1 && ps.nslots == 1 && c % 64 == 0 && (act == 0 || act == 1) &&
            (rt->conv_variant == 5 ||
             (rt->conv_variant < 0 && f >= 128 && ceil_div(f, 256) * ceil_div((long)n * ((oh * ow + 7) / 8 * 8), 256) * 16 >= rt->num_cu * 3))) {
            const int st = launch_conv_pw_gemm(rt, dtype, ps.o, w, bias, res, y, n, c, (long)oh * ow, f, act);
            if (st >= 0)
                return st;
        }
    }
    if (tap_want) {
        bool safe = true;
        if (!split) {
            // a tap moves a 16-byte run by up to one row + one pixel: the bytes in front of and behind X it then reaches (masked away,
            // but fetched) must be readable memory. True inside an arena of infini_rocm_alloc (256 bytes of slack on both sides) and
            // for a tensor in the middle of a caller's block; a tensor at the very edge of its allocation takes the other kernels.
            const long reach = ((long)ow + 1) * 2 + 16;
            hipDeviceptr_t base = nullptr;
            size_t size = 0;
            if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)x) != hipSuccess) {
                (void)hipGetLastError();
                safe = false;
            } else {
                const char *lo = (const char *)x - reach, *hi = (const char *)x + p.plane_elems * 2 + reach;
                safe = lo >= (const char *)base && hi <= (const char *)base + size;
            }
        }
        if (safe) {
            const int st = launch_conv_tap_gemm(rt, dtype, p.x, p.w, bias, y, n, c, oh, ow, h, wd, sh, p.plane_elems, f, act, tap_split,
                                                tap_slab_bytes ? ws + ws_slab_off : nullptr, tap_slab_bytes);
            if (st >= 0)
                return st;
        }
        if (split && rt->conv_variant == 7) // (the tap-shifted kernel below reads the planes through p.slot: any order serves it)
            rt->last_conv_route = "tap_shifted";
    }
    const bool bf = dtype == INFINI_DT_BF16;
    static const int pw_on = getenv("IROCM_CONV_PW") ? atoi(getenv("IROCM_CONV_PW")) : 1; // tuning hook: 0 = off
    if (pw_on && r == 1 && s == 1 && ph == 0 && pw == 0 && c % 64 == 0 && c <= (pw_on == 2 ? 256 : 128) && f > 64 && p.hw % 2 == 0)
        return bf ? launch_pw<Bf16Traits>(rt, p) : launch_pw<F16Traits>(rt, p);
    if (rowtap)
        return bf ? launch_s1<Bf16Traits, 1, 4, 32, true>(rt, p) : launch_s1<F16Traits, 1, 4, 32, true>(rt, p);
    // unit-stride "same" R x S (the 3x3 layers): input patch resident in LDS, every tap an aligned row offset
    static const int patch_on = getenv("IROCM_CONV_PATCH") ? atoi(getenv("IROCM_CONV_PATCH")) : 1; // tuning hook: 0 = off
    if (patch_on && r * s > 1 && r * s <= 32 && !split && dh == 1 && dw == 1 && oh == h && ow == wd && 2 * ph == r - 1 &&
        2 * pw == s - 1 && c % 32 == 0 && rt->conv_variant != 4) { // variant 4: the tap-shifted kernel (A/B)
        const int halo8 = (ph * wd + pw + 7) & ~7;
        // 128 f x 128 slots. (The 64 f x 256 slots form of the same kernel, <1, 4>, was measured on ResNet's C64 -> F64
        // 56x56 layers — two channel blocks, 18 taps per workgroup: 119 us against 96 us for the tap-shifted kernel, whose
        // 64 x 256 x 32 tile has no patch / transpose prologue to amortise — and is not instantiated.)
        // 128 f x 256 slots on 8 waves (one workgroup per CU, three weight stages) when that still fills most of the chip: every
        // weight tile streamed from L2 then serves twice the slots. C128 28x28 60.6 -> 58.3 us, C256 14x14 55.1 -> 51.6; C512 7x7
        // (100 workgroups) 73.8 -> 83.2: stays on the 4-wave form. IROCM_CONV_PATCH_WIDE = 0 / 1 forces either (A/B); conv variant 6 forces the wide form (tests, tune()).
        static const int patch_wide = getenv("IROCM_CONV_PATCH_WIDE") ? atoi(getenv("IROCM_CONV_PATCH_WIDE")) : -1;
        const bool wide_fills = ceil_div(f, 128) * ceil_div(p.ncols, 256) * 10 >= (long)rt->num_cu * 7;
        if (f > 64 && 2 * halo8 <= 128 && (patch_wide == 1 || rt->conv_variant == 6 || (patch_wide < 0 && wide_fills)))
            return bf ? launch_patch<Bf16Traits, 2, 4>(rt, p, halo8, 256 + 2 * halo8)
                      : launch_patch<F16Traits, 2, 4>(rt, p, halo8, 256 + 2 * halo8);
        if (f > 64 && 2 * halo8 <= 128)
            return bf ? launch_patch<Bf16Traits, 2, 2>(rt, p, halo8, 128 + 2 * halo8)
                      : launch_patch<F16Traits, 2, 2>(rt, p, halo8, 128 + 2 * halo8);
        // F <= 64, C <= 64: the whole weight tensor resident in LDS, persistent workgroups over 256-slot tiles (conv_resident)
        static const int resident_on = getenv("IROCM_CONV_RESIDENT") ? atoi(getenv("IROCM_CONV_RESIDENT")) : 1; // A/B hook
        if (resident_on && f <= 64 && c <= 64 && p.hw % 8 == 0 && 2 * halo8 <= 128 && (((uintptr_t)x) & 15) == 0 && !res &&
            r == 3 && s == 3 && ph == 1 && pw == 1 && (c == 32 || c == 64) && p.wide_epilogue == 2 && rt->conv_variant != 4)
            return bf ? launch_resident<Bf16Traits>(rt, p, halo8) : launch_resident<F16Traits>(rt, p, halo8);
    }
    if (f <= 64)
        return bf ? launch_s1<Bf16Traits, 1, 4, 32>(rt, p) : launch_s1<F16Traits, 1, 4, 32>(rt, p);
    static const int cfg = getenv("IROCM_CONV_CFG") ? atoi(getenv("IROCM_CONV_CFG")) : 0; // tuning hook
    if (c % 64 != 0 || cfg == 1)
        return bf ? launch_s1<Bf16Traits, 2, 2, 32>(rt, p) : launch_s1<F16Traits, 2, 2, 32>(rt, p);
    return bf ? launch_s1<Bf16Traits, 2, 2, 64>(rt, p) : launch_s1<F16Traits, 2, 2, 64>(rt, p);
}

} // namespace irocm
