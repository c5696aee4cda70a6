// HBM-bound row operators for gfx950: Softmax, LayerNormalization, RMSNorm.
//
// Softmax    replaces softmax_kernel      (reference: src/kernels/cuda/softmax.cu:242-404)
// LayerNorm  replaces LaynormKernel       (reference: src/kernels/cuda/layer_norm.cu:339-558)
// RMSNorm    replaces rmsnorm_kernel      (reference: src/kernels/cuda/rms_norm.cu:35-110)
//
// Design (MI355X): every row is read from HBM exactly once and written once — algorithmic bytes
// 2 * numel * sizeof(T). A row of up to 64 * 16 * VPT elements lives in the registers
// of ONE wave64 (16-byte loads per lane, `VEC` elements each); max / sum / mean / variance are
// wave reductions through DPP/ds_swizzle shuffles (no LDS, no barrier). Longer rows use one
// 256-thread block per row with the row cached in registers and a 4-wave LDS combine.
// Strided (inner > 1) softmax maps lanes along the contiguous inner dimension instead, so
// accesses stay coalesced, and walks the softmax axis with the online max/sum recurrence of the
// reference (softmax.cu:8-17).
// All arithmetic is fp32 regardless of the storage dtype.
#include "common.h"
#include <cstdlib>

namespace irocm {

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int VEC = 4; // 16 B
    __device__ static inline float ld(const float *p) { return *p; }
    __device__ static inline void st(float *p, float v) { *p = v; }
};
template <> struct Elem<__half> {
    static constexpr int VEC = 8;
    __device__ static inline float ld(const __half *p) { return __half2float(*p); }
    __device__ static inline void st(__half *p, float v) { *p = __float2half_rn(v); }
};
template <> struct Elem<__hip_bfloat16> {
    static constexpr int VEC = 8;
    __device__ static inline float ld(const __hip_bfloat16 *p) { return __bfloat162float(*p); }
    __device__ static inline void st(__hip_bfloat16 *p, float v) { *p = __float2bfloat16(v); }
};

template <typename T, int N> struct alignas(sizeof(T) * N) Pack {
    T v[N];
};

// Wave64 all-reduce on the DPP cross-lane paths (no LDS round trip: __shfl_xor is ds_bpermute, ~100+ cycles a step):
// quad swaps, half-row / row mirrors give every lane of a 16-lane row the row total; row_bcast15 / row_bcast31 chain
// the four rows so lane 63 holds the wave total, which v_readlane broadcasts. The order of the additions is fixed.
template <int CTRL, int ROW_MASK> __device__ inline float dpp_move(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xf, false));
}
__device__ inline float wave_bcast63(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ inline float wave_max(float v) {
    v = fmaxf(v, dpp_move<0xB1, 0xf>(v, v));  // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_move<0x4E, 0xf>(v, v));  // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_move<0x141, 0xf>(v, v)); // row_half_mirror
    v = fmaxf(v, dpp_move<0x140, 0xf>(v, v)); // row_mirror
    v = fmaxf(v, dpp_move<0x142, 0xa>(v, v)); // row_bcast15 into rows 1, 3
    v = fmaxf(v, dpp_move<0x143, 0xc>(v, v)); // row_bcast31 into rows 2, 3
    return wave_bcast63(v);
}
__device__ inline float wave_sum(float v) {
    v += dpp_move<0xB1, 0xf>(0.f, v);
    v += dpp_move<0x4E, 0xf>(0.f, v);
    v += dpp_move<0x141, 0xf>(0.f, v);
    v += dpp_move<0x140, 0xf>(0.f, v);
    v += dpp_move<0x142, 0xa>(0.f, v);
    v += dpp_move<0x143, 0xc>(0.f, v);
    return wave_bcast63(v);
}

// One expression for every normalisation kernel (explicit, non-contractable steps), so that the fused Add -> Norm
// kernel and the stand-alone kernels round identically.
__device__ inline float norm_apply(float x, float mu, float rstd, float g, float b) {
    return fmaf(__fmul_rn(__fsub_rn(x, mu), rstd), g, b);
}

__device__ inline float norm_apply_centered(float d, float rstd, float g, float b) { // d = x - mu
    return fmaf(__fmul_rn(d, rstd), g, b);
}

// ------------------------------------------------------------------------------------------------
// Contiguous rows (inner == 1). One wave handles ROWS rows; a row is CHUNKS 16-byte chunks per lane,
// kept PACKED in registers (storage dtype) and converted on the fly in every pass: for 16-bit types
// this halves the register footprint, which is what buys the occupancy (bytes in flight) the HBM
// pipe needs. Handles any dimsize <= 64 * VEC * CHUNKS. ALIGNED: row bases 16-B aligned and
// dimsize % VEC == 0 (pure vector loads); otherwise element loads into the same register image.
// ------------------------------------------------------------------------------------------------
template <typename T, int CHUNKS> struct RowRegs {
    Pack<T, Elem<T>::VEC> c[CHUNKS];
};

template <typename T, int CHUNKS, bool ALIGNED>
__device__ inline void load_row(const T *row, int n, int lane, RowRegs<T, CHUNKS> &r, float fill) {
    constexpr int VEC = Elem<T>::VEC;
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        const int base = (c * 64 + lane) * VEC;
        if (ALIGNED) {
            if (base < n) {
                r.c[c] = *reinterpret_cast<const Pack<T, VEC> *>(row + base);
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    Elem<T>::st(&r.c[c].v[j], fill);
            }
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                if (base + j < n)
                    r.c[c].v[j] = row[base + j];
                else
                    Elem<T>::st(&r.c[c].v[j], fill);
            }
        }
    }
}

template <typename T, int CHUNKS, bool ALIGNED>
__device__ inline void store_row(T *row, int n, int lane, const RowRegs<T, CHUNKS> &r) {
    constexpr int VEC = Elem<T>::VEC;
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        const int base = (c * 64 + lane) * VEC;
        if (ALIGNED) {
            if (base < n)
                *reinterpret_cast<Pack<T, VEC> *>(row + base) = r.c[c];
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                if (base + j < n)
                    row[base + j] = r.c[c].v[j];
        }
    }
}

// ROWS rows per wave per iteration; waves are persistent (grid-stride over row groups) and software
// pipelined: the loads of group g+1 are issued before the arithmetic of group g, so every wave has
// loads in flight during its reductions instead of only at its start.
template <typename T, int CHUNKS, bool ALIGNED, int ROWS>
__global__ __launch_bounds__(256) void softmax_wave_kernel(const T *__restrict__ x, T *__restrict__ y,
                                                           long rows, int n) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = threadIdx.x & 63;
    const long ngroups = (rows + ROWS - 1) / ROWS;
    const long stride = (long)gridDim.x * 4;
    long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= ngroups)
        return;
    RowRegs<T, CHUNKS> cur[ROWS], nxt[ROWS];
    auto load_group = [&](RowRegs<T, CHUNKS>(&dst)[ROWS], long grp) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            long row = grp * ROWS + r;
            row = row < rows ? row : rows - 1; // tail rows recompute the last row (never stored)
            load_row<T, CHUNKS, ALIGNED>(x + row * n, n, lane, dst[r], -INFINITY);
        }
    };
    load_group(cur, g);
    for (; g < ngroups; g += stride) {
        const bool more = g + stride < ngroups;
        if (more)
            load_group(nxt, g + stride);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            float m = -INFINITY;
#pragma unroll
            for (int c = 0; c < CHUNKS; ++c)
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    m = fmaxf(m, Elem<T>::ld(&cur[r].c[c].v[j]));
            m = wave_max(m);
            float e[CHUNKS * VEC];
            float s = 0.f;
            // 16-bit storage: e = 2^(x log2e - m log2e) as ONE fma + the raw v_exp_f32 (no denormal-range fix-up: a weight
            // below 2^-126 is 0 in f16 / bf16 anyway); fp32 keeps expf (1e-6-class accuracy against the oracle)
            const float ml2 = m * 1.4426950408889634f;
#pragma unroll
            for (int c = 0; c < CHUNKS; ++c)
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float xv = Elem<T>::ld(&cur[r].c[c].v[j]);
                    const float ev = sizeof(T) == 4 ? expf(xv - m) : __builtin_amdgcn_exp2f(fmaf(xv, 1.4426950408889634f, -ml2));
                    e[c * VEC + j] = ev;
                    s += ev;
                }
            s = wave_sum(s);
            const float inv = 1.0f / s;
#pragma unroll
            for (int c = 0; c < CHUNKS; ++c)
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    Elem<T>::st(&cur[r].c[c].v[j], e[c * VEC + j] * inv);
            const long row = g * ROWS + r;
            if (row < rows)
                store_row<T, CHUNKS, ALIGNED>(y + row * n, n, lane, cur[r]);
        }
        if (more) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
                cur[r] = nxt[r];
        }
    }
}

// Long contiguous rows: one 256-thread block per row, grid-stride over rows, three passes over a
// row kept in L2/MALL when it does not fit registers (rare: dimsize > 8192).
template <typename T>
__global__ __launch_bounds__(256) void softmax_block_kernel(const T *__restrict__ x, T *__restrict__ y,
                                                            long rows, long n) {
    __shared__ float red[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const T *xr = x + row * n;
        T *yr = y + row * n;
        float m = -INFINITY, s = 0.f;
        for (long i = t; i < n; i += 256) { // online max/sum (reference softmax.cu:8-17)
            const float v = Elem<T>::ld(xr + i);
            const float nm = fmaxf(m, v);
            s = s * __expf(m - nm) + __expf(v - nm);
            m = nm;
        }
        float gm = wave_max(m);
        s *= __expf(m - gm);
        s = wave_sum(s);
        if (lane == 0)
            red[w] = gm;
        __syncthreads();
        const float bm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
        s *= __expf(gm - bm);
        if (lane == 0)
            red[w] = s;
        __syncthreads();
        const float bs = red[0] + red[1] + red[2] + red[3];
        __syncthreads();
        const float inv = 1.0f / bs;
        for (long i = t; i < n; i += 256)
            Elem<T>::st(yr + i, __expf(Elem<T>::ld(xr + i) - bm) * inv);
    }
}

// Strided softmax: tensor [outer, dimsize, inner], inner > 1. One thread per (outer, inner)
// column; consecutive lanes take consecutive inner positions (coalesced); online recurrence.
template <typename T>
__global__ __launch_bounds__(256) void softmax_strided_kernel(const T *__restrict__ x, T *__restrict__ y,
                                                              long outer, long dimsize, long inner) {
    const long total = outer * inner;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const long o = idx / inner, i = idx % inner;
        const T *xp = x + o * dimsize * inner + i;
        T *yp = y + o * dimsize * inner + i;
        float m = -INFINITY, s = 0.f;
        for (long d = 0; d < dimsize; ++d) {
            const float v = Elem<T>::ld(xp + d * inner);
            const float nm = fmaxf(m, v);
            s = s * __expf(m - nm) + __expf(v - nm);
            m = nm;
        }
        const float inv = 1.0f / s;
        for (long d = 0; d < dimsize; ++d)
            Elem<T>::st(yp + d * inner, __expf(Elem<T>::ld(xp + d * inner) - m) * inv);
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm / RMSNorm, one wave per row (row in registers), two-pass mean/variance in registers
// (numerically the "centered" form: var = mean((x - mu)^2)), fp32 throughout.
// ------------------------------------------------------------------------------------------------
template <typename T, int CHUNKS, bool ALIGNED, bool RMS, int ROWS>
__global__ __launch_bounds__(256) void norm_wave_kernel(const T *__restrict__ x, const T *__restrict__ scale,
                                                        const T *__restrict__ bias, T *__restrict__ y,
                                                        long rows, int n, int scale_size, int bias_size,
                                                        float eps) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = threadIdx.x & 63;
    const long ngroups = (rows + ROWS - 1) / ROWS;
    const long stride = (long)gridDim.x * 4;
    long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= ngroups)
        return;
    RowRegs<T, CHUNKS> cur[ROWS], nxt[ROWS];
    auto load_group = [&](RowRegs<T, CHUNKS>(&dst)[ROWS], long grp) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            long row = grp * ROWS + r;
            row = row < rows ? row : rows - 1;
            load_row<T, CHUNKS, ALIGNED>(x + row * n, n, lane, dst[r], 0.f);
        }
    };
    load_group(cur, g);
    // scale / bias: per-element (size n) or scalar (size 1); loaded once per (persistent) wave, packed
    RowRegs<T, CHUNKS> sc, bs;
    float s0 = 1.f, b0 = 0.f;
    const bool sc_vec = scale_size != 1, bs_vec = bias != nullptr && bias_size != 1;
    if (sc_vec)
        load_row<T, CHUNKS, ALIGNED>(scale, n, lane, sc, 0.f);
    else
        s0 = Elem<T>::ld(scale);
    if (bs_vec)
        load_row<T, CHUNKS, ALIGNED>(bias, n, lane, bs, 0.f);
    else if (bias != nullptr)
        b0 = Elem<T>::ld(bias);
    const float fn = (float)n;
    for (; g < ngroups; g += stride) {
        const bool more = g + stride < ngroups;
        if (more)
            load_group(nxt, g + stride);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            float mu = 0.f;
            if (!RMS) {
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < CHUNKS; ++c)
#pragma unroll
                    for (int j = 0; j < VEC; ++j)
                        s += Elem<T>::ld(&cur[r].c[c].v[j]); // padding lanes hold 0
                mu = wave_sum(s) / fn; // correctly rounded: integer-valued rows give exact means
            }
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < CHUNKS; ++c)
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const int col = (c * 64 + lane) * VEC + j;
                    const float d = (col < n) ? Elem<T>::ld(&cur[r].c[c].v[j]) - mu : 0.f;
                    q = fmaf(d, d, q);
                }
            const float rstd = 1.0f / sqrtf(wave_sum(q) / fn + eps);
#pragma unroll
            for (int c = 0; c < CHUNKS; ++c)
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float gg = sc_vec ? Elem<T>::ld(&sc.c[c].v[j]) : s0;
                    const float bb = bs_vec ? Elem<T>::ld(&bs.c[c].v[j]) : b0;
                    Elem<T>::st(&cur[r].c[c].v[j], norm_apply(Elem<T>::ld(&cur[r].c[c].v[j]), mu, rstd, gg, bb));
                }
            const long row = g * ROWS + r;
            if (row < rows)
                store_row<T, CHUNKS, ALIGNED>(y + row * n, n, lane, cur[r]);
        }
        if (more) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
                cur[r] = nxt[r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Aligned contiguous rows, the fast path (BERT / Llama hidden sizes). Measured on the first version of the wave
// kernel: LayerNorm f16 16384 x 768 ran at ~1470 SIMD cycles per row whatever the byte count -- VALU-issue bound
// (per-element selects and conversions, ds_bpermute reductions, a vmcnt(0) behind the prefetch branch), not HBM bound.
// This kernel
//  * addresses a row through a per-row BUFFER descriptor (the row base is wave-uniform): lanes past the row end read 0
//    and their stores are dropped by the hardware range check, so loads / stores carry no predicate and no branch, and
//    the next row's prefetch is cancelled by a zero-sized descriptor instead of a branch (a branch join would force
//    vmcnt(0) and serialise prefetch and arithmetic);
//  * picks the chunk width B (16 or 8 bytes per lane) that leaves the fewest idle lanes: n = 768 f16 is 3 chunks of 8 B
//    (12 elements per lane) instead of 2 chunks of 16 B with half the lanes idle in the second;
//  * does the arithmetic on float2 (v_pk_add / v_pk_mul / v_pk_fma_f32), converts once, keeps scale / bias in fp32
//    registers for the life of the persistent wave, masks the variance only in the last chunk;
//  * reduces with the DPP wave_sum above.
// Same value semantics as norm_apply(): fp32, centered variance, (x - mu) * rstd * g + b with one fma.
// ------------------------------------------------------------------------------------------------
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int rw_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int rw_u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 rw_f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 rw_bf16x2 __attribute__((ext_vector_type(2)));

typedef float rw_f32x4 __attribute__((ext_vector_type(4)));

// One chunk = the B bytes a lane loads at once, kept as the loaded dword vector; Chunk converts it to / from float2 pairs
// with whole-vector operations only (shuffles, convertvector). Going through scalar dword arrays here made the
// optimizer (SLP) emit a wrong fp32 kernel -- it computed half of the outputs and duplicated them.
template <int B> struct ChunkVec;
template <> struct ChunkVec<16> { using type = rw_u32x4; };
template <> struct ChunkVec<8> { using type = rw_u32x2; };

template <typename T> __device__ inline f32x2_t pair_up(unsigned u);
template <> __device__ inline f32x2_t pair_up<__half>(unsigned u) {
    return __builtin_convertvector(__builtin_bit_cast(rw_f16x2, u), f32x2_t);
}
template <> __device__ inline f32x2_t pair_up<__hip_bfloat16>(unsigned u) {
    return f32x2_t{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
}
template <typename T> __device__ inline unsigned pair_down(f32x2_t v);
template <> __device__ inline unsigned pair_down<__half>(f32x2_t v) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, rw_f16x2)); // v_cvt_pk_f16_f32, RNE
}
template <> __device__ inline unsigned pair_down<__hip_bfloat16>(f32x2_t v) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, rw_bf16x2)); // v_cvt_pk_bf16_f32, RNE
}

template <typename T, int B> struct Chunk {
    using U = typename ChunkVec<B>::type;
    static constexpr int P = B / (2 * (int)sizeof(T)); // float2 pairs per chunk
    __device__ static inline void up(U v, f32x2_t *dst) {
        if constexpr (sizeof(T) == 4 && B == 16) {
            const rw_f32x4 f = __builtin_bit_cast(rw_f32x4, v);
            dst[0] = __builtin_shufflevector(f, f, 0, 1);
            dst[1] = __builtin_shufflevector(f, f, 2, 3);
        } else if constexpr (sizeof(T) == 4) {
            dst[0] = __builtin_bit_cast(f32x2_t, v);
        } else {
#pragma unroll
            for (int k = 0; k < P; ++k)
                dst[k] = pair_up<T>(v[k]);
        }
    }
    __device__ static inline U down(const f32x2_t *src) {
        if constexpr (sizeof(T) == 4 && B == 16) {
            return __builtin_bit_cast(U, __builtin_shufflevector(src[0], src[1], 0, 1, 2, 3));
        } else if constexpr (sizeof(T) == 4) {
            return __builtin_bit_cast(U, src[0]);
        } else {
            U r;
#pragma unroll
            for (int k = 0; k < P; ++k)
                r[k] = pair_down<T>(src[k]);
            return r;
        }
    }
};

template <int B, int CHUNKS>
__device__ __forceinline__ void rows_load(typename ChunkVec<B>::type (&dst)[CHUNKS], const void *base, int nbytes, int voff0) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, nbytes, 0x00020000);
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        if constexpr (B == 16)
            dst[c] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff0 + c * 64 * B, 0, 0);
        else
            dst[c] = __builtin_amdgcn_raw_buffer_load_b64(rs, voff0 + c * 64 * B, 0, 0);
    }
}
template <int B, int CHUNKS>
__device__ __forceinline__ void rows_store(const typename ChunkVec<B>::type (&src)[CHUNKS], void *base, int nbytes, int voff0) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, nbytes, 0x00020000);
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        if constexpr (B == 16)
            __builtin_amdgcn_raw_buffer_store_b128(src[c], rs, voff0 + c * 64 * B, 0, 0);
        else
            __builtin_amdgcn_raw_buffer_store_b64(src[c], rs, voff0 + c * 64 * B, 0, 0);
    }
}

// 1 / sqrt(a): v_rsq_f32 (1 ulp) + one Newton step -> < 1 ulp, 4 instructions instead of the ~25 of 1.0f / sqrtf(a)
__device__ inline float inv_sqrt(float a) {
    const float y = __builtin_amdgcn_rsqf(a);
    const float h = 0.5f * a * y;
    return fmaf(y, fmaf(-h, y, 0.5f), y);
}

// ADD = 1: the row is a + b (x2 = b), rounded to the storage type exactly as a separate Add kernel would store it, so
// Norm(Add(a, b)) fused here equals the two-kernel chain bit for bit (this is the Add -> LayerNorm / RMSNorm fusion).
// ADD = 2: the row is (a + pre) + b with `pre` one row vector of n elements (the bias of the linear layer that produced
// a: MatMul -> Add(bias) -> Add(residual) -> Norm as the ONNX front-end emits a transformer's output projections), each
// sum rounded like its own Add kernel would have stored it.
template <typename T, int B, int CHUNKS, bool RMS, int ROWS, int ADD>
__global__ __launch_bounds__(256) void norm_rows_kernel(const T *__restrict__ x, const T *__restrict__ x2,
                                                        const T *__restrict__ pre,
                                                        const T *__restrict__ scale, const T *__restrict__ bias,
                                                        T *__restrict__ y, long rows, int n, int scale_size,
                                                        int bias_size, float eps) {
    using CK = Chunk<T, B>;
    using U = typename CK::U;
    constexpr int PL = CK::P;          // float2 pairs in one chunk
    constexpr int NP = PL * CHUNKS;    // pairs of one row held by a lane
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long ngroups = (rows + ROWS - 1) / ROWS;
    const long stride = (long)gridDim.x * 4;
    long g = (long)blockIdx.x * 4 + w;
    if (g >= ngroups)
        return;
    const int row_bytes = n * (int)sizeof(T);
    const int voff0 = lane * B;
    const bool vlast = voff0 + (CHUNKS - 1) * 64 * B < row_bytes;
    constexpr int R2 = ADD ? ROWS : 1;
    U cur[ROWS][CHUNKS], nxt[ROWS][CHUNKS], cur2[R2][CHUNKS], nxt2[R2][CHUNKS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const long row = g * ROWS + r;
        rows_load<B, CHUNKS>(cur[r], x + (row < rows ? row : rows - 1) * n, row_bytes, voff0);
        if constexpr (ADD)
            rows_load<B, CHUNKS>(cur2[r], x2 + (row < rows ? row : rows - 1) * n, row_bytes, voff0);
    }
    f32x2_t gs[NP], bs[NP], ps[ADD == 2 ? NP : 1];
    {
        U tmp[CHUNKS];
        if constexpr (ADD == 2) {
            rows_load<B, CHUNKS>(tmp, pre, row_bytes, voff0);
#pragma unroll
            for (int c = 0; c < CHUNKS; ++c)
                CK::up(tmp[c], &ps[c * PL]);
        }
        if (scale_size != 1) {
            rows_load<B, CHUNKS>(tmp, scale, row_bytes, voff0);
#pragma unroll
            for (int c = 0; c < CHUNKS; ++c)
                CK::up(tmp[c], &gs[c * PL]);
        } else {
            const float s0 = Elem<T>::ld(scale);
#pragma unroll
            for (int i = 0; i < NP; ++i)
                gs[i] = f32x2_t{s0, s0};
        }
        if (bias != nullptr && bias_size != 1) {
            rows_load<B, CHUNKS>(tmp, bias, row_bytes, voff0);
#pragma unroll
            for (int c = 0; c < CHUNKS; ++c)
                CK::up(tmp[c], &bs[c * PL]);
        } else {
            const float b0 = bias != nullptr ? Elem<T>::ld(bias) : 0.f;
#pragma unroll
            for (int i = 0; i < NP; ++i)
                bs[i] = f32x2_t{b0, b0};
        }
    }
    const float fn = (float)n, inv_n = 1.0f / fn;
    for (; g < ngroups; g += stride) {
        const long gn = g + stride;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { // prefetch; past the end the descriptor is empty: no traffic, no branch
            const long row = gn * ROWS + r;
            const bool live = gn < ngroups && row < rows;
            rows_load<B, CHUNKS>(nxt[r], x + (live ? row : 0) * n, live ? row_bytes : 0, voff0);
            if constexpr (ADD)
                rows_load<B, CHUNKS>(nxt2[r], x2 + (live ? row : 0) * n, live ? row_bytes : 0, voff0);
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            f32x2_t xf[NP];
#pragma unroll
            for (int c = 0; c < CHUNKS; ++c)
                CK::up(cur[r][c], &xf[c * PL]);
            if constexpr (ADD == 2) {
#pragma unroll
                for (int i = 0; i < NP; ++i)
                    xf[i] += ps[i];
                if constexpr (sizeof(T) == 2) {
#pragma unroll
                    for (int i = 0; i < NP; ++i)
                        xf[i] = pair_up<T>(pair_down<T>(xf[i]));
                }
            }
            if constexpr (ADD) {
                f32x2_t bf[NP];
#pragma unroll
                for (int c = 0; c < CHUNKS; ++c)
                    CK::up(cur2[r][c], &bf[c * PL]);
#pragma unroll
                for (int i = 0; i < NP; ++i)
                    xf[i] += bf[i];
                if constexpr (sizeof(T) == 2) { // round the sum to the storage type, like the Add kernel's store
#pragma unroll
                    for (int i = 0; i < NP; ++i)
                        xf[i] = pair_up<T>(pair_down<T>(xf[i]));
                }
            }
            f32x2_t mu2 = {0.f, 0.f};
            if (!RMS) {
                f32x2_t s2 = xf[0];
#pragma unroll
                for (int i = 1; i < NP; ++i)
                    s2 += xf[i]; // lanes past the row end hold 0
                const float mu = wave_sum(s2.x + s2.y) / fn; // correctly rounded: integer-valued rows give exact means
                mu2 = f32x2_t{mu, mu};
            }
            f32x2_t q2 = {0.f, 0.f}, ql = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                xf[i] -= mu2; // the centered value is what the output needs too
                if (i < NP - PL)
                    q2 = __builtin_elementwise_fma(xf[i], xf[i], q2);
                else
                    ql = __builtin_elementwise_fma(xf[i], xf[i], ql);
            }
            const float qlast = (RMS || vlast) ? ql.x + ql.y : 0.f; // padding lanes would add mu^2
            const float rstd = inv_sqrt(fmaf(wave_sum(q2.x + q2.y + qlast), inv_n, eps));
            const f32x2_t r2 = {rstd, rstd};
            U out[CHUNKS];
#pragma unroll
            for (int c = 0; c < CHUNKS; ++c) {
                f32x2_t o[PL];
#pragma unroll
                for (int k = 0; k < PL; ++k) // same roundings as norm_apply(): (x - mu) * rstd, then one fma
                    o[k] = __builtin_elementwise_fma(xf[c * PL + k] * r2, gs[c * PL + k], bs[c * PL + k]);
                out[c] = CK::down(o);
            }
            const long row = g * ROWS + r;
            rows_store<B, CHUNKS>(out, y + (row < rows ? row : 0) * n, row < rows ? row_bytes : 0, voff0);
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int c = 0; c < CHUNKS; ++c) {
                cur[r][c] = nxt[r][c];
                if constexpr (ADD)
                    cur2[r][c] = nxt2[r][c];
            }
    }
}


// Long rows: block per row, row re-read from L2.
template <typename T, bool RMS>
__global__ __launch_bounds__(256) void norm_block_kernel(const T *__restrict__ x, const T *__restrict__ scale,
                                                         const T *__restrict__ bias, T *__restrict__ y,
                                                         long rows, long n, int scale_size, int bias_size,
                                                         float eps) {
    __shared__ float red[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const T *xr = x + row * n;
        T *yr = y + row * n;
        float mu = 0.f;
        if (!RMS) {
            float s = 0.f;
            for (long i = t; i < n; i += 256)
                s += Elem<T>::ld(xr + i);
            s = wave_sum(s);
            if (lane == 0)
                red[w] = s;
            __syncthreads();
            mu = (red[0] + red[1] + red[2] + red[3]) / (float)n;
            __syncthreads();
        }
        float q = 0.f;
        for (long i = t; i < n; i += 256) {
            const float d = Elem<T>::ld(xr + i) - mu;
            q = fmaf(d, d, q);
        }
        q = wave_sum(q);
        if (lane == 0)
            red[w] = q;
        __syncthreads();
        const float rstd = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)n + eps);
        __syncthreads();
        for (long i = t; i < n; i += 256) {
            const float s = Elem<T>::ld(scale + (scale_size == 1 ? 0 : i));
            const float b = bias ? Elem<T>::ld(bias + (bias_size == 1 ? 0 : i)) : 0.f;
            Elem<T>::st(yr + i, norm_apply(Elem<T>::ld(xr + i), mu, rstd, s, b));
        }
    }
}

// Long aligned rows (up to 256 * VEC * CH elements): one 256-thread block per row, the row kept PACKED in registers
// (CH 16-byte chunks per thread), 16-byte loads / stores, one read and one write of HBM per element; the two
// reductions go wave-shuffle -> 4-entry LDS combine. (The scalar three-pass kernels above ran a 16384 x 4096 f16
// LayerNorm at 1.9 TB/s.)
__device__ inline float block_sum(float v, float *red, int lane, int w) {
    v = wave_sum(v);
    if (lane == 0)
        red[w] = v;
    __syncthreads();
    const float r = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return r;
}
__device__ inline float block_max(float v, float *red, int lane, int w) {
    v = wave_max(v);
    if (lane == 0)
        red[w] = v;
    __syncthreads();
    const float r = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    return r;
}

// ADD = 1: the row is a + b (x2 = b); ADD = 2: (a + pre) + b — each sum rounded to T like its own Add kernel would have
// stored it (the same contract as norm_rows_kernel; round 4: rows beyond 4 KiB, e.g. the 4096-wide rows of a Llama block,
// used to take a separate Add pass — 0.34 of the HBM peak for the pair).
template <typename T, int CH, bool RMS, int ADD>
__global__ __launch_bounds__(256) void norm_blockreg_kernel(const T *__restrict__ x, const T *x2, const T *__restrict__ pre,
                                                            const T *__restrict__ scale, const T *__restrict__ bias, T *y,
                                                            long rows, int n, int scale_size, int bias_size, float eps) {
    constexpr int VEC = Elem<T>::VEC;
    using P = Pack<T, VEC>;
    __shared__ float red[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const T *xr = x + row * (long)n;
        T *yr = y + row * (long)n;
        P c[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int base = (i * 256 + t) * VEC;
            if (base < n)
                c[i] = *reinterpret_cast<const P *>(xr + base);
        }
        if constexpr (ADD != 0) {
            const T *br = x2 + row * (long)n; // (may be y itself: every element is read before the row is written)
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int base = (i * 256 + t) * VEC;
                if (base < n) {
                    const P b2 = *reinterpret_cast<const P *>(br + base);
                    P pv;
                    if constexpr (ADD == 2)
                        pv = *reinterpret_cast<const P *>(pre + base);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        float a = Elem<T>::ld(&c[i].v[j]);
                        if constexpr (ADD == 2) {
                            T mid;
                            Elem<T>::st(&mid, a + Elem<T>::ld(&pv.v[j]));
                            a = Elem<T>::ld(&mid);
                        }
                        Elem<T>::st(&c[i].v[j], a + Elem<T>::ld(&b2.v[j]));
                    }
                }
            }
        }
        float mu = 0.f;
        if (!RMS) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < CH; ++i)
                if ((i * 256 + t) * VEC < n) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j)
                        s += Elem<T>::ld(&c[i].v[j]);
                }
            mu = block_sum(s, red, lane, w) / (float)n;
        }
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if ((i * 256 + t) * VEC < n) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float d = Elem<T>::ld(&c[i].v[j]) - mu;
                    q = fmaf(d, d, q);
                }
            }
        const float rstd = 1.0f / sqrtf(block_sum(q, red, lane, w) / (float)n + eps);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int base = (i * 256 + t) * VEC;
            if (base < n) {
                P sc, bi, o;
                if (scale_size != 1)
                    sc = *reinterpret_cast<const P *>(scale + base);
                if (bias && bias_size != 1)
                    bi = *reinterpret_cast<const P *>(bias + base);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float sv = scale_size == 1 ? Elem<T>::ld(scale) : Elem<T>::ld(&sc.v[j]);
                    const float bv = bias ? (bias_size == 1 ? Elem<T>::ld(bias) : Elem<T>::ld(&bi.v[j])) : 0.f;
                    Elem<T>::st(&o.v[j], norm_apply(Elem<T>::ld(&c[i].v[j]), mu, rstd, sv, bv));
                }
                *reinterpret_cast<P *>(yr + base) = o;
            }
        }
    }
}

template <typename T, int CH>
__global__ __launch_bounds__(256) void softmax_blockreg_kernel(const T *__restrict__ x, T *__restrict__ y, long rows,
                                                               int n) {
    constexpr int VEC = Elem<T>::VEC;
    using P = Pack<T, VEC>;
    __shared__ float red[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const T *xr = x + row * (long)n;
        T *yr = y + row * (long)n;
        P c[CH];
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int base = (i * 256 + t) * VEC;
            if (base < n) {
                c[i] = *reinterpret_cast<const P *>(xr + base);
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    m = fmaxf(m, Elem<T>::ld(&c[i].v[j]));
            }
        }
        m = block_max(m, red, lane, w);
        float e[CH][VEC];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if ((i * 256 + t) * VEC < n) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    e[i][j] = __expf(Elem<T>::ld(&c[i].v[j]) - m);
                    s += e[i][j];
                }
            }
        const float inv = 1.0f / block_sum(s, red, lane, w);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int base = (i * 256 + t) * VEC;
            if (base < n) {
                P o;
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    Elem<T>::st(&o.v[j], e[i][j] * inv);
                *reinterpret_cast<P *>(yr + base) = o;
            }
        }
    }
}

static inline bool is_aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }
static inline int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
// The persistent grid of ONE kernel: as many 256-thread workgroups per CU as are RESIDENT at once (asked from the HIP runtime once
// per kernel; at most 8). With a fixed eight per CU a kernel of 86 registers (five resident) ran 5 + 3: the last three workgroups of
// every CU started when the first five had finished all their rows — the same tail nnops.hip's short-row reduction had (0.56 -> 0.68).
template <auto Kern> static inline unsigned pgrid_k(int64_t blocks, int num_cu) {
    static const int per_cu = [] {
        const int forced = env_int("IROCM_ROWOPS_BLOCKS_PER_CU", 0); // tuning hook
        if (forced > 0)
            return forced;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, Kern, 256, 0) != hipSuccess || nb < 1)
            nb = 8;
        return nb < 8 ? nb : 8;
    }();
    const int64_t cap = (int64_t)num_cu * per_cu;
    return (unsigned)(blocks < cap ? (blocks < 1 ? 1 : blocks) : cap);
}

// Launch of the fast row path (norm_rows_kernel); false when the rows do not qualify (unaligned, or > 4 KiB).
template <typename T, bool RMS, int ADD>
static bool launch_norm_rows(infiniRocmRuntime_t rt, const T *x, const T *x2, const T *pre, const T *scale, const T *bias, T *y,
                             int64_t outer, int64_t n, int64_t scale_size, int64_t bias_size, float eps) {
    constexpr int VEC = Elem<T>::VEC;
    const int64_t row_bytes = n * (int64_t)sizeof(T);
    const bool al = is_aligned16(x) && (!ADD || is_aligned16(x2)) && (ADD != 2 || is_aligned16(pre)) && is_aligned16(y) &&
                    is_aligned16(scale) && (bias == nullptr || is_aligned16(bias)) && (n % VEC == 0);
    if (!al || row_bytes > 4096 || outer == 0)
        return false;
    // chunk width: the one that leaves the fewest idle lanes (768 f16 = 3 x 8 B, not 2 x 16 B with half a chunk idle)
    const int c16 = (int)ceil_div(row_bytes, (int64_t)1024), c8 = (int)ceil_div(row_bytes, (int64_t)512);
    const bool use8 = (c8 == 1 || c8 == 3) && (c8 * 512 - row_bytes) < (c16 * 1024 - row_bytes);
    static const int rpw_env = env_int("IROCM_NORM_RPW", 0); // tuning hook
    static const int per_cu = env_int("IROCM_NORM_BLOCKS_PER_CU", 4); // 16 persistent waves per CU, rows pipelined
    const int rpw = (!ADD && rpw_env == 2) ? 2 : 1;
    const int64_t blocks = ceil_div(outer, (int64_t)4 * rpw), cap = (int64_t)rt->num_cu * per_cu;
    const dim3 grid((unsigned)(blocks < cap ? blocks : cap));
#define ROWS_GO(B, C, R)                                                                                        \
    hipLaunchKernelGGL((norm_rows_kernel<T, B, C, RMS, R, ADD>), grid, dim3(256), 0, rt->stream, x, x2, pre, scale, bias, y, \
                       (long)outer, (int)n, (int)scale_size, (int)bias_size, eps)
#define ROWS_R(B, C)                                                                                            \
    do {                                                                                                        \
        if constexpr (!ADD) {                                                                                   \
            if (rpw == 2) ROWS_GO(B, C, 2); else ROWS_GO(B, C, 1);                                               \
        } else {                                                                                                \
            ROWS_GO(B, C, 1);                                                                                   \
        }                                                                                                       \
    } while (0)
    if (use8) {
        if (c8 == 1) ROWS_R(8, 1); else ROWS_R(8, 3);
    } else {
        switch (c16) {
        case 1: ROWS_R(16, 1); break;
        case 2: ROWS_R(16, 2); break;
        case 3: ROWS_R(16, 3); break;
        default: ROWS_R(16, 4); break;
        }
    }
#undef ROWS_R
#undef ROWS_GO
    return true;
}

template <typename T>
static int softmax_dispatch(infiniRocmRuntime_t rt, const T *x, T *y, int64_t outer, int64_t dimsize,
                            int64_t inner) {
    constexpr int VEC = Elem<T>::VEC;
    if (inner == 1) {
        const bool al = is_aligned16(x) && is_aligned16(y) && (dimsize % VEC == 0);
        const int64_t per_chunk = 64 * VEC;
        const int chunks = (int)ceil_div(dimsize, per_chunk);
        // rows per wave: keep ~4 KiB of loads in flight per wave
        const int64_t row_bytes = dimsize * (int64_t)sizeof(T);
        static const int rpw8 = getenv("IROCM_SOFTMAX_RPW8") ? atoi(getenv("IROCM_SOFTMAX_RPW8")) : 0; // tuning hook
        const int rpw = (!al || outer < 4096) ? 1 : (row_bytes <= 1024 ? (rpw8 ? 8 : 4) : (row_bytes <= 3072 ? 2 : 1));
#define SM_GO(C, A, R)                                                                             \
    hipLaunchKernelGGL((softmax_wave_kernel<T, C, A, R>), dim3(pgrid_k<softmax_wave_kernel<T, C, A, R>>(ceil_div(outer, 4 * R), rt->num_cu)), \
                       dim3(256), 0, rt->stream, x, y, (long)outer, (int)dimsize)
        if (chunks <= 1) {
            if (!al) SM_GO(1, false, 1);
            else if (rpw == 8) SM_GO(1, true, 8);
            else if (rpw == 4) SM_GO(1, true, 4);
            else if (rpw == 2) SM_GO(1, true, 2);
            else SM_GO(1, true, 1);
        } else if (chunks <= 2) {
            if (!al) SM_GO(2, false, 1);
            else if (rpw == 4) SM_GO(2, true, 4);
            else if (rpw == 2) SM_GO(2, true, 2);
            else SM_GO(2, true, 1);
        } else if (chunks <= 4) {
            if (al) SM_GO(4, true, 1); else SM_GO(4, false, 1);
        } else if (chunks <= 8) {
            if (al) SM_GO(8, true, 1); else SM_GO(8, false, 1);
        }
        else {
            const unsigned g = (unsigned)(outer < 8192 ? outer : 8192);
            const int bch = (int)ceil_div(dimsize, (int64_t)256 * VEC);
            if (al && bch <= 4)
                hipLaunchKernelGGL((softmax_blockreg_kernel<T, 4>), dim3(g), dim3(256), 0, rt->stream, x, y, (long)outer,
                                   (int)dimsize);
            else if (al && bch <= 8)
                hipLaunchKernelGGL((softmax_blockreg_kernel<T, 8>), dim3(g), dim3(256), 0, rt->stream, x, y, (long)outer,
                                   (int)dimsize);
            else
                hipLaunchKernelGGL((softmax_block_kernel<T>), dim3(g), dim3(256), 0, rt->stream, x, y,
                                   (long)outer, (long)dimsize);
        }
#undef SM_GO
    } else {
        const int64_t total = outer * inner;
        const unsigned g = (unsigned)(ceil_div(total, 256) < 16384 ? ceil_div(total, 256) : 16384);
        hipLaunchKernelGGL((softmax_strided_kernel<T>), dim3(g), dim3(256), 0, rt->stream, x, y,
                           (long)outer, (long)dimsize, (long)inner);
    }
    IROCM_LAUNCH_CHECK("softmax");
    return INFINI_ROCM_OK;
}

template <typename T, bool RMS>
static int norm_dispatch(infiniRocmRuntime_t rt, const T *x, const T *scale, const T *bias, T *y,
                         int64_t outer, int64_t n, int64_t scale_size, int64_t bias_size, float eps) {
    constexpr int VEC = Elem<T>::VEC;
    const bool al = is_aligned16(x) && is_aligned16(y) && is_aligned16(scale) &&
                    (bias == nullptr || is_aligned16(bias)) && (n % VEC == 0);
    const int chunks = (int)ceil_div(n, (int64_t)64 * VEC);
    if (launch_norm_rows<T, RMS, 0>(rt, x, (const T *)nullptr, (const T *)nullptr, scale, bias, y, outer, n, scale_size, bias_size, eps)) {
        IROCM_LAUNCH_CHECK("norm");
        return INFINI_ROCM_OK;
    }
#define NORM_GO(C, A, R)                                                                           \
    hipLaunchKernelGGL((norm_wave_kernel<T, C, A, RMS, R>), dim3(pgrid_k<norm_wave_kernel<T, C, A, RMS, R>>(ceil_div(outer, 4 * R), rt->num_cu)), \
                       dim3(256), 0, rt->stream, x, scale, bias, y, (long)outer, (int)n,           \
                       (int)scale_size, (int)bias_size, eps)
    if (chunks <= 1) {
        NORM_GO(1, false, 1);
    } else if (chunks <= 2) {
        NORM_GO(2, false, 1);
    } else if (chunks <= 4) {
        NORM_GO(4, false, 1);
    } else {
        const unsigned g = (unsigned)(outer < 8192 ? outer : 8192);
        const int bch = (int)ceil_div(n, (int64_t)256 * VEC); // 16-byte chunks per thread of a block-resident row
#define NORM_BR(C)                                                                                 \
    hipLaunchKernelGGL((norm_blockreg_kernel<T, C, RMS, 0>), dim3(g), dim3(256), 0, rt->stream, x, (const T *)nullptr,   \
                       (const T *)nullptr, scale, bias, y, (long)outer, (int)n, (int)scale_size, (int)bias_size, eps)
        if (al && bch <= 2) NORM_BR(2);
        else if (al && bch <= 4) NORM_BR(4);
        else if (al && bch <= 8) NORM_BR(8);
        else
            hipLaunchKernelGGL((norm_block_kernel<T, RMS>), dim3(g), dim3(256), 0, rt->stream, x, scale,
                               bias, y, (long)outer, (long)n, (int)scale_size, (int)bias_size, eps);
#undef NORM_BR
    }
#undef NORM_GO
    IROCM_LAUNCH_CHECK("norm");
    return INFINI_ROCM_OK;
}

// Returns INFINI_ROCM_UNSUPPORTED (without setting an error) when the shape is outside the fused kernel's reach.
template <typename T, bool RMS>
static int add_norm_dispatch(infiniRocmRuntime_t rt, const T *a, const T *b, const T *pre, const T *scale, const T *bias, T *y,
                             int64_t outer, int64_t n, int64_t scale_size, int64_t bias_size, float eps) {
    const bool ok = pre ? launch_norm_rows<T, RMS, 2>(rt, a, b, pre, scale, bias, y, outer, n, scale_size, bias_size, eps)
                        : launch_norm_rows<T, RMS, 1>(rt, a, b, pre, scale, bias, y, outer, n, scale_size, bias_size, eps);
    if (!ok) {
        // rows beyond the wave-resident kernel's 4 KiB: block-resident rows (up to 256 threads x 8 chunks x 16 B = 32 KiB)
        constexpr int VEC = Elem<T>::VEC;
        const bool al = is_aligned16(a) && is_aligned16(b) && (!pre || is_aligned16(pre)) && is_aligned16(y) && is_aligned16(scale) &&
                        (bias == nullptr || is_aligned16(bias)) && (n % VEC == 0);
        const int bch = (int)ceil_div(n, (int64_t)256 * VEC);
        if (!al || bch > 8 || outer == 0)
            return INFINI_ROCM_UNSUPPORTED; // the caller falls back to Add + Norm
        const unsigned g = (unsigned)(outer < 8192 ? outer : 8192);
#define ADD_BR(C, A)                                                                                                  \
    hipLaunchKernelGGL((norm_blockreg_kernel<T, C, RMS, A>), dim3(g), dim3(256), 0, rt->stream, a, b, pre, scale, bias, y, \
                       (long)outer, (int)n, (int)scale_size, (int)bias_size, eps)
#define ADD_BRC(A)                                                                                                    \
    do {                                                                                                              \
        if (bch <= 2) ADD_BR(2, A); else if (bch <= 4) ADD_BR(4, A); else ADD_BR(8, A);                               \
    } while (0)
        if (pre) ADD_BRC(2); else ADD_BRC(1);
#undef ADD_BRC
#undef ADD_BR
    }
    IROCM_LAUNCH_CHECK("add_norm");
    return INFINI_ROCM_OK;
}

} // namespace irocm

using namespace irocm;

extern "C" {

int infini_rocm_softmax(infiniRocmRuntime_t rt, int dtype, const void *x, void *y, int64_t outer,
                        int64_t dimsize, int64_t inner) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(outer >= 0 && dimsize >= 0 && inner >= 0, "softmax: negative extent");
    if (outer == 0 || dimsize == 0 || inner == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(x && y, "softmax: NULL tensor");
    IROCM_CHECK_ARG(dimsize < (1ll << 31), "softmax: dimsize too large");
    switch (dtype) {
    case INFINI_DT_F32:
        return softmax_dispatch<float>(rt, (const float *)x, (float *)y, outer, dimsize, inner);
    case INFINI_DT_F16:
        return softmax_dispatch<__half>(rt, (const __half *)x, (__half *)y, outer, dimsize, inner);
    case INFINI_DT_BF16:
        return softmax_dispatch<__hip_bfloat16>(rt, (const __hip_bfloat16 *)x, (__hip_bfloat16 *)y,
                                                outer, dimsize, inner);
    default:
        IROCM_FAIL(INFINI_ROCM_UNSUPPORTED, "softmax: unsupported dtype %s", dtype_name(dtype));
    }
}

int infini_rocm_layer_norm(infiniRocmRuntime_t rt, int dtype, const void *x, const void *scale,
                           const void *bias, void *y, int64_t outer, int64_t norm_size,
                           int64_t scale_size, int64_t bias_size, float eps) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(outer >= 0 && norm_size >= 0, "layer_norm: negative extent");
    if (outer == 0 || norm_size == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(x && y && scale, "layer_norm: NULL tensor");
    IROCM_CHECK_ARG(scale_size == 1 || scale_size == norm_size,
                    "layer_norm: scale has %lld elements, expected 1 or %lld", (long long)scale_size,
                    (long long)norm_size);
    IROCM_CHECK_ARG(bias == nullptr || bias_size == 1 || bias_size == norm_size,
                    "layer_norm: bias has %lld elements, expected 1 or %lld", (long long)bias_size,
                    (long long)norm_size);
    IROCM_CHECK_ARG(norm_size < (1ll << 31), "layer_norm: norm_size too large");
    switch (dtype) {
    case INFINI_DT_F32:
        return norm_dispatch<float, false>(rt, (const float *)x, (const float *)scale,
                                           (const float *)bias, (float *)y, outer, norm_size,
                                           scale_size, bias_size, eps);
    case INFINI_DT_F16:
        return norm_dispatch<__half, false>(rt, (const __half *)x, (const __half *)scale,
                                            (const __half *)bias, (__half *)y, outer, norm_size,
                                            scale_size, bias_size, eps);
    case INFINI_DT_BF16:
        return norm_dispatch<__hip_bfloat16, false>(
            rt, (const __hip_bfloat16 *)x, (const __hip_bfloat16 *)scale, (const __hip_bfloat16 *)bias,
            (__hip_bfloat16 *)y, outer, norm_size, scale_size, bias_size, eps);
    default:
        IROCM_FAIL(INFINI_ROCM_UNSUPPORTED, "layer_norm: unsupported dtype %s", dtype_name(dtype));
    }
}

int infini_rocm_add_norm(infiniRocmRuntime_t rt, int dtype, int rms, const void *a, const void *b, const void *scale,
                         const void *bias, void *y, int64_t outer, int64_t norm_size, int64_t scale_size,
                         int64_t bias_size, float eps) {
    return infini_rocm_bias_add_norm(rt, dtype, rms, a, nullptr, b, scale, bias, y, outer, norm_size, scale_size, bias_size, eps);
}

int infini_rocm_bias_add_norm(infiniRocmRuntime_t rt, int dtype, int rms, const void *a, const void *pre, const void *b,
                              const void *scale, const void *bias, void *y, int64_t outer, int64_t norm_size,
                              int64_t scale_size, int64_t bias_size, float eps) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(outer >= 0 && norm_size >= 0 && norm_size < (1ll << 31), "add_norm: bad extent");
    if (outer == 0 || norm_size == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(a && b && y && scale, "add_norm: NULL tensor");
    IROCM_CHECK_ARG(scale_size == 1 || scale_size == norm_size, "add_norm: bad scale size");
    IROCM_CHECK_ARG(bias == nullptr || bias_size == 1 || bias_size == norm_size, "add_norm: bad bias size");
    int st = INFINI_ROCM_UNSUPPORTED;
#define GO(T)                                                                                                   \
    st = rms ? add_norm_dispatch<T, true>(rt, (const T *)a, (const T *)b, (const T *)pre, (const T *)scale, (const T *)bias, \
                                          (T *)y, outer, norm_size, scale_size, bias_size, eps)                \
             : add_norm_dispatch<T, false>(rt, (const T *)a, (const T *)b, (const T *)pre, (const T *)scale, (const T *)bias, \
                                           (T *)y, outer, norm_size, scale_size, bias_size, eps)
    switch (dtype) {
    case INFINI_DT_F32: GO(float); break;
    case INFINI_DT_F16: GO(__half); break;
    case INFINI_DT_BF16: GO(__hip_bfloat16); break;
    default: IROCM_FAIL(INFINI_ROCM_UNSUPPORTED, "add_norm: unsupported dtype %s", dtype_name(dtype));
    }
#undef GO
    if (st != INFINI_ROCM_UNSUPPORTED)
        return st;
    // outside the fused kernel's reach (long or unaligned rows): the operator chain, in place over y. y may BE a or b (the
    // planner allows the row-wise in-place forms): with a row bias the two Adds run as ONE element-wise pass
    // (round(round(a + pre) + b), every element read before it is written), never as y = a + pre followed by y + b, which
    // would read an overwritten b when y aliases it.
    if (pre) {
        st = infini_rocm_bias_residual(rt, dtype, a, pre, b, y, outer, norm_size, 1, 0);
    } else {
        const int64_t shape[2] = {outer, norm_size}, str[2] = {norm_size, 1};
        st = infini_rocm_binary(rt, INFINI_BIN_ADD, dtype, a, b, y, 2, shape, str, str);
    }
    if (st != INFINI_ROCM_OK)
        return st;
    return rms ? infini_rocm_rms_norm(rt, dtype, y, scale, y, outer, norm_size, eps)
               : infini_rocm_layer_norm(rt, dtype, y, scale, bias, y, outer, norm_size, scale_size, bias_size, eps);
}

int infini_rocm_rms_norm(infiniRocmRuntime_t rt, int dtype, const void *x, const void *w, void *y,
                         int64_t outer, int64_t norm_size, float eps) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(outer >= 0 && norm_size >= 0, "rms_norm: negative extent");
    if (outer == 0 || norm_size == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(x && y && w, "rms_norm: NULL tensor");
    IROCM_CHECK_ARG(norm_size < (1ll << 31), "rms_norm: norm_size too large");
    switch (dtype) {
    case INFINI_DT_F32:
        return norm_dispatch<float, true>(rt, (const float *)x, (const float *)w, nullptr, (float *)y,
                                          outer, norm_size, norm_size, 0, eps);
    case INFINI_DT_F16:
        return norm_dispatch<__half, true>(rt, (const __half *)x, (const __half *)w, nullptr,
                                           (__half *)y, outer, norm_size, norm_size, 0, eps);
    case INFINI_DT_BF16:
        return norm_dispatch<__hip_bfloat16, true>(rt, (const __hip_bfloat16 *)x,
                                                   (const __hip_bfloat16 *)w, nullptr,
                                                   (__hip_bfloat16 *)y, outer, norm_size, norm_size, 0,
                                                   eps);
    default:
        IROCM_FAIL(INFINI_ROCM_UNSUPPORTED, "rms_norm: unsupported dtype %s", dtype_name(dtype));
    }
}

} // extern "C"
