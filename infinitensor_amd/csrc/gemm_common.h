// Shared device helpers for the MFMA GEMM / implicit-GEMM conv kernels (gfx950 only).
#pragma once
#include "common.h"

namespace irocm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

#define IROCM_LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define IROCM_GLB_PTR(p) ((const __attribute__((address_space(1))) void *)(p))

struct GemmArgs {
    const void *a, *b, *bias;
    void *c;
    int m, n, k, batch;
    // element strides: A(i,kk) = a[ib*a_bs + i*a_rs + kk*a_cs]; B(kk,j) = b[ib*b_bs + kk*b_rs + j*b_cs]
    long a_rs, a_cs, a_bs;
    long b_rs, b_cs, b_bs;
    long bias_b, bias_m, bias_n;
    long c_bs; // element stride between the batches' [m x n] output blocks (m * n unless the caller groups separate tensors)
    int act;
    int tiles_m, tiles_n;
    // split-K (gemm256 only): `splitk` workgroups share one output tile, each sums a slice of K into its own fp32
    // plane of `partial` [splitk][batch][m][n]; splitk_reduce adds the planes, the bias and the activation
    int splitk;
    float *partial;
    int epi16;         // gemm256: 16-byte stores in the epilogue (pair exchange between lane groups)
    const void *zeros; // >= 16 zero bytes in device memory: source of K-tail chunks past k (fast128)
    // head-split store (0 = off): C[row][col] of one [m x n] problem goes to a [m / hs_s][n / hs_d][hs_s][hs_d] tensor,
    // i.e. the MatMul -> Reshape([B, S, H, D]) -> Transpose(0, 2, 1, 3) chain of a transformer's q / k / v projections
    // written by the GEMM itself. hs_d % 8 == 0 so a 16-byte store never straddles two heads.
    int hs_s, hs_d;
    // conv mode of the persistent kernels (gemm256p_kernel.h, CONV): a pointwise convolution Y[img][f][pix] = W[f][c] X[img][c][pix]
    // as ONE GEMM whose columns are pixel SLOTS img * cv_hwp + pix (cv_hwp = the plane rounded up to 8, so a 16-byte run never
    // straddles two images): m = F, k = C, n = images * cv_hwp; b = X, c = Y (NCHW), bias = one value per ROW (filter),
    // cv_res = the residual added before the activation (NCHW like Y, or nullptr). 0 = off.
    int cv_hw = 0, cv_hwp = 0;
    const void *cv_res = nullptr;
    unsigned cv_hwp_m = 0;     // floor(2^32 / cv_hwp): slot -> (image, pixel) by multiply-high (udivmod_m below)
    // Tap mode of the conv mode (gemm256p_kernel.h, CONV = 3; round 5): an R x S = 3 x 3 convolution as ONE GEMM with K = 9 C — K-tile
    // index -> (channel block of 64, tap), TAP INNER (the nine taps of a channel block re-read nearly the same activation bytes:
    // L2 hits). a = weights re-packed [tap][F][C] (conv_repack_w), b = the activation (unit stride: X itself; stride 2: the four
    // de-interleaved phase planes in slot order py * 2 + px). The B tile of tap (r, s) is the pointwise tile moved by a constant
    // byte offset rowoff[r] + coloff[s]; zero padding is a per-lane AND on the B fragments (one validity bit per (slot, tap)).
    int cv_taps = 0;                 // 0: pointwise; 9: 3 x 3
    int cv_ow = 0;                   // output row length: slot -> (oy, ox)
    unsigned cv_ow_m = 0;            // floor(2^32 / cv_ow)
    int cv_ylo = 0, cv_yhi = 0;      // tap row r = 0 lies inside the image for oy >= cv_ylo, r = 2 for oy < cv_yhi (r = 1 always)
    int cv_xlo = 0, cv_xhi = 0;      // columns alike
    int cv_b0 = 0;                   // byte offset of tap (0, 0) relative to b (negative for a unit-stride layer: -(W + 1) * 2)
    int cv_ds01 = 0, cv_ds12 = 0;    // byte-offset steps of B: s 0 -> 1, s 1 -> 2
    int cv_dr01 = 0, cv_dr12 = 0;    //   r 0 -> 1, r 1 -> 2 (s back to 0 included)
    int cv_dcb = 0;                  //   tap (2, 2) -> tap (0, 0) of the next channel block
    int cv_atap = 0;                 // bytes between two taps' weight images: F * C * 2
    unsigned cv_res_bytes = 0; // bytes of the residual tensor (images * F * hw * 2): the range of its buffer descriptor, computed on the host
                          // (computed in the kernel — a division — hipcc kept it in vector registers and wrapped every buffer load of
                          // the residual in a readfirstlane waterfall loop)
};

// element offset of C(row, col) inside one batch's [m x n] block
__device__ __forceinline__ long c_off(const GemmArgs &p, long row, long col) {
    if (p.hs_d == 0)
        return row * p.n + col;
    const long b = row / p.hs_s, s = row - b * p.hs_s, h = col / p.hs_d, d = col - h * p.hs_d;
    return ((b * (p.n / p.hs_d) + h) * p.hs_s + s) * p.hs_d + d;
}

// 16-bit element traits: how to feed v_mfma_f32_16x16x32_{bf16,f16} and convert on store.
struct Bf16Traits {
    static constexpr int kDType = INFINI_DT_BF16;
    __device__ static inline f32x4 mfma(s16x8_t a, s16x8_t b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                       __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
    // round-to-nearest-even (same rounding torch / the oracle use): one v_cvt_pk_bf16_f32 on gfx950
    __device__ static inline unsigned short from_f32(float f) {
        return __builtin_bit_cast(unsigned short, (__bf16)f);
    }
    __device__ static inline float to_f32(unsigned short h) {
        return __builtin_bit_cast(float, ((unsigned int)h) << 16);
    }
    // two values -> one dword {lo, hi}: ONE v_cvt_pk_bf16_f32 (from_f32 twice + shift + or is four VALU slots)
    __device__ static inline unsigned pack2(float lo, float hi) {
        typedef float f32x2_ __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){lo, hi}, bf16x2_));
    }
};

struct F16Traits {
    static constexpr int kDType = INFINI_DT_F16;
    __device__ static inline f32x4 mfma(s16x8_t a, s16x8_t b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                      __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
    __device__ static inline unsigned short from_f32(float f) {
        _Float16 h = (_Float16)f; // v_cvt_f16_f32: RNE
        return __builtin_bit_cast(unsigned short, h);
    }
    __device__ static inline float to_f32(unsigned short h) {
        return (float)__builtin_bit_cast(_Float16, h);
    }
    // two values -> one dword {lo, hi}: one v_cvt_pk_f16_f32 (RNE; gfx950)
    __device__ static inline unsigned pack2(float lo, float hi) {
        typedef float f32x2_ __attribute__((ext_vector_type(2)));
        typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){lo, hi}, f16x2_));
    }
};

// Gelu (erf form, unary.cu) with erf by Abramowitz-Stegun 7.1.26 (abs error <= 1.5e-7, far below a 16-bit output's
// rounding): ~18 VALU slots instead of erff's ~40, so it can ride in a GEMM epilogue. 1 +- erf is formed without
// cancellation: 1 - erf(z) = poly(t) e^{-z^2}.
__device__ static inline float gelu_erf_as(float v) {
    // 0.5 v (1 + erf(v / sqrt 2)) = max(v, 0) - 0.5 |v| c  with  c = 1 - erf(|v| / sqrt 2) = poly(t) t e^{-v^2 / 2}: one form for
    // both signs, the two scale factors folded into constants
    // (p / sqrt 2; sqrt(log2 e / 2) so that exp2 takes -(k |v|)^2 directly): 14 plain VALU + rcp + exp2 per element.
    // |v| is clamped at 1e4 (c is exactly 0 long before) so that +inf gives inf, not inf * 0; the relu part carries a NaN.
    const float a = fminf(fabsf(v), 1.0e4f);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, a, 1.0f));
    float q = fmaf(1.061405429f, t, -1.453152027f);
    q = fmaf(q, t, 1.421413741f);
    q = fmaf(q, t, -0.284496736f);
    q = fmaf(q, t, 0.254829592f);
    const float zz = a * 0.84932180028801904272f; // sqrt(log2(e) / 2)
    const float c = (q * t) * __builtin_amdgcn_exp2f(-(zz * zz));
    return fmaf(-0.5f, a * c, v < 0.f ? 0.f : v);
}

// Fused epilogue activation (reference ActType: include/core/common.h — None/Relu/Sigmoid/Tanh).
__device__ static inline float apply_act(float v, int act) {
    switch (act) {
    case 1: return v > 0.f ? v : 0.f;
    case 2: return 1.f / (1.f + __expf(-v));
    case 3: return tanhf(v);
    case 4: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); // Gelu (erf form, unary.cu) — fusion only
    case 5: return gelu_poly(v);
    default: return v;
    }
}

// Bijective XCD-aware remap of a linear workgroup id: consecutive ids returned to one XCD
// (dispatch is round-robin over the 8 XCDs: block b runs on XCD b % 8 — speed only).
// n / d and n % d for n < 2^32 with m = min(floor(2^32 / d), 2^32 - 1): the multiply-high quotient is short by at most one.
// (A division by a run-time value is ~45 instructions; the persistent kernels did 9-17 of them per tile.)
__device__ __forceinline__ void udivmod_m(unsigned n, unsigned d, unsigned m, unsigned &q, unsigned &r) {
    q = __umulhi(n, m);
    r = n - q * d;
    if (r >= d) {
        ++q;
        r -= d;
    }
}
static inline unsigned udiv_magic(unsigned long long d) { return d <= 1 ? 0xffffffffu : (unsigned)((1ull << 32) / d); }

__device__ static inline unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg / kNumXcd, r = nwg % kNumXcd;
    const unsigned xcd = bid % kNumXcd, idx = bid / kNumXcd;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// ---- 128-row LDS-DMA operand tiles shared by gemm.hip and conv.hip -------------------------------
namespace f128 {
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2; // 16 KiB per operand tile

// K-major operand tile: image [128 rows][64 k] (128 B rows), physical 16-B chunk c' of row r holds
// logical chunk c' ^ ((r >> 1) & 7): ds_read_b128 lane groups then hit 16 distinct 16-B slots.
// `kend`: chunks starting at k >= kend are fetched from `zeros` instead (K tail of a matrix whose k is a multiple of 8
// but not of BK); pass kend = INT_MAX-like and zeros = nullptr-safe value when there is no tail.
__device__ inline void stage_kmajor(const unsigned short *base, long ld, int row0, int rows, int k0,
                                    char *lds_tile, int w, int lane, int kend = 0x7fffffff,
                                    const unsigned short *zeros = nullptr) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = w * 4 + i;
        const int r = piece * 8 + (lane >> 3);
        const int c_log = (lane & 7) ^ ((r >> 1) & 7);
        int gr = row0 + r;
        gr = gr < rows ? gr : rows - 1;
        const unsigned short *src = base + (long)gr * ld + k0 + c_log * 8;
        if (k0 + c_log * 8 >= kend)
            src = zeros;
        __builtin_amdgcn_global_load_lds(IROCM_GLB_PTR(src), IROCM_LDS_PTR(lds_tile + piece * 1024), 16, 0, 0);
    }
}

// M/N-major operand tile: image [64 k][128 cols] (256 B rows), 32-B chunk index XORed with
// f(k) = (k & 3) | ((k >> 3) & 1) << 2 so the 8 k-rows one tr-read half-wave touches are spread
// over the whole 256-B bank row.
__device__ inline int mn_f(int kr) { return (kr & 3) | (((kr >> 3) & 1) << 2); }

__device__ inline void stage_mnmajor(const unsigned short *base, long ld, int col0, int cols, int k0,
                                     char *lds_tile, int w, int lane, int kend = 0x7fffffff,
                                     const unsigned short *zeros = nullptr) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = w * 4 + i;
        const int kr = piece * 4 + (lane >> 4);
        const int c_log = (lane & 15) ^ (mn_f(kr) << 1);
        int gc = col0 + c_log * 8;
        gc = gc <= cols - 8 ? gc : cols - 8;
        const unsigned short *src = base + (long)(k0 + kr) * ld + gc;
        if (k0 + kr >= kend)
            src = zeros;
        __builtin_amdgcn_global_load_lds(IROCM_GLB_PTR(src), IROCM_LDS_PTR(lds_tile + piece * 1024), 16, 0, 0);
    }
}

// MFMA 16x16x32 operand fragment: lane l holds 8 consecutive k (ks*32 + (l>>4)*8 ..) of row/col
// R0 + (l & 15).
__device__ inline s16x8_t frag_kmajor(const char *lds_tile, int R0, int ks, int lane) {
    const int r = R0 + (lane & 15);
    const int c = (ks * 4 + (lane >> 4)) ^ ((r >> 1) & 7);
    return *(const s16x8_t *)(lds_tile + r * 128 + c * 16);
}

__device__ inline s16x8_t frag_mnmajor(const char *lds_tile, int C0, int ks, int lane) {
    // Two ds_read_b64_tr_b16: in each 16-lane group, lane p supplies the address of 4 consecutive
    // columns (p & 3) * 4 of k-row (p >> 2) and receives column p of that 4x16 block.
    const int p = lane & 15;
    s16x4_t h[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int kr = ks * 32 + (lane >> 4) * 8 + hh * 4 + (p >> 2);
        const int col = C0 + (p & 3) * 4;
        const int c16 = (col >> 3) ^ (mn_f(kr) << 1);
        const char *addr = lds_tile + kr * 256 + c16 * 16 + (col & 4) * 2;
        h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t *)(addr));
    }
    return s16x8_t{h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
}
} // namespace f128

} // namespace irocm
