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

__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o, 64);
    return v;
}

// One expression for every normalisation kernel (explicit, non-contractable steps), so that the fused Add -> Norm
// kernel and the stand-alone kernels round identically.
__device__ inline float norm_apply(float x, float mu, float rstd, float g, float b) {
    return fmaf(__fmul_rn(__fsub_rn(x, mu), rstd), g, b);
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
#pragma unroll
            for (int c = 0; c < CHUNKS; ++c)
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float d = Elem<T>::ld(&cur[r].c[c].v[j]) - m;
                    const float ev = sizeof(T) == 4 ? expf(d) : __expf(d);
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

template <typename T, int CH, bool RMS>
__global__ __launch_bounds__(256) void norm_blockreg_kernel(const T *__restrict__ x, const T *__restrict__ scale,
                                                            const T *__restrict__ bias, T *__restrict__ y, long rows,
                                                            int n, int scale_size, int bias_size, float eps) {
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

// Add -> LayerNorm / RMSNorm in one pass (the residual join in front of every transformer normalisation):
// y = norm(round_T(a + b)). The sum is rounded to the storage type before the statistics, exactly like the unfused
// Add -> Norm chain, so the result equals it up to fp32 rounding ties of the final store. One wave per row, rows of up to 64 * VEC * CHUNKS elements,
// 16-byte loads of both operands, grid-stride over rows.
template <typename T, int CHUNKS, bool RMS>
__global__ __launch_bounds__(256) void add_norm_wave_kernel(const T *__restrict__ a, const T *__restrict__ b,
                                                            const T *__restrict__ scale, const T *__restrict__ bias,
                                                            T *__restrict__ y, long rows, int n, int scale_size,
                                                            int bias_size, float eps) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = threadIdx.x & 63;
    const float fn = (float)n;
    for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
        RowRegs<T, CHUNKS> ra, rb;
        load_row<T, CHUNKS, true>(a + row * n, n, lane, ra, 0.f);
        load_row<T, CHUNKS, true>(b + row * n, n, lane, rb, 0.f);
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c)
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                Elem<T>::st(&ra.c[c].v[j], Elem<T>::ld(&ra.c[c].v[j]) + Elem<T>::ld(&rb.c[c].v[j]));
        float mu = 0.f;
        if (!RMS) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < CHUNKS; ++c)
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    s += Elem<T>::ld(&ra.c[c].v[j]);
            mu = wave_sum(s) / fn;
        }
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c)
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const int col = (c * 64 + lane) * VEC + j;
                const float d = (col < n) ? Elem<T>::ld(&ra.c[c].v[j]) - mu : 0.f;
                q = fmaf(d, d, q);
            }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / fn + eps);
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) {
            const int base = (c * 64 + lane) * VEC;
            if (base < n) {
                Pack<T, VEC> sc, bs;
                if (scale_size != 1)
                    sc = *reinterpret_cast<const Pack<T, VEC> *>(scale + base);
                if (bias && bias_size != 1)
                    bs = *reinterpret_cast<const Pack<T, VEC> *>(bias + base);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float gg = scale_size == 1 ? Elem<T>::ld(scale) : Elem<T>::ld(&sc.v[j]);
                    const float bb = bias ? (bias_size == 1 ? Elem<T>::ld(bias) : Elem<T>::ld(&bs.v[j])) : 0.f;
                    Elem<T>::st(&ra.c[c].v[j], norm_apply(Elem<T>::ld(&ra.c[c].v[j]), mu, rstd, gg, bb));
                }
            }
        }
        store_row<T, CHUNKS, true>(y + row * n, n, lane, ra);
    }
}

static inline bool is_aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }
// persistent grid: at most 8 blocks (32 waves) per CU
static inline int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
static inline unsigned pgrid(int64_t blocks, int num_cu) {
    static const int per_cu = env_int("IROCM_ROWOPS_BLOCKS_PER_CU", 8); // tuning hook
    const int64_t cap = (int64_t)num_cu * per_cu;
    return (unsigned)(blocks < cap ? (blocks < 1 ? 1 : blocks) : cap);
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
        const int rpw = (!al || outer < 4096) ? 1 : (row_bytes <= 1024 ? 4 : (row_bytes <= 3072 ? 2 : 1));
#define SM_GO(C, A, R)                                                                             \
    hipLaunchKernelGGL((softmax_wave_kernel<T, C, A, R>), dim3(pgrid(ceil_div(outer, 4 * R), rt->num_cu)), \
                       dim3(256), 0, rt->stream, x, y, (long)outer, (int)dimsize)
        if (chunks <= 1) {
            if (!al) SM_GO(1, false, 1);
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
    const int64_t row_bytes = n * (int64_t)sizeof(T);
    int rpw = (!al || outer < 4096) ? 1 : (row_bytes <= 1024 ? 4 : (row_bytes <= 3072 ? 2 : 1));
    static const int rpw_env = env_int("IROCM_NORM_RPW", 0); // tuning hook
    if (rpw_env && al)
        rpw = rpw_env;
#define NORM_GO(C, A, R)                                                                           \
    hipLaunchKernelGGL((norm_wave_kernel<T, C, A, RMS, R>), dim3(pgrid(ceil_div(outer, 4 * R), rt->num_cu)), \
                       dim3(256), 0, rt->stream, x, scale, bias, y, (long)outer, (int)n,           \
                       (int)scale_size, (int)bias_size, eps)
    if (chunks <= 1) {
        if (!al) NORM_GO(1, false, 1);
        else if (rpw == 4) NORM_GO(1, true, 4);
        else if (rpw == 2) NORM_GO(1, true, 2);
        else NORM_GO(1, true, 1);
    } else if (chunks <= 2) {
        if (!al) NORM_GO(2, false, 1);
        else if (rpw == 4) NORM_GO(2, true, 4);
        else if (rpw == 2) NORM_GO(2, true, 2);
        else NORM_GO(2, true, 1);
    } else if (chunks <= 4) {
        if (al) NORM_GO(4, true, 1); else NORM_GO(4, false, 1);
    } else {
        const unsigned g = (unsigned)(outer < 8192 ? outer : 8192);
        const int bch = (int)ceil_div(n, (int64_t)256 * VEC); // 16-byte chunks per thread of a block-resident row
#define NORM_BR(C)                                                                                 \
    hipLaunchKernelGGL((norm_blockreg_kernel<T, C, RMS>), dim3(g), dim3(256), 0, rt->stream, x, scale, bias, y, \
                       (long)outer, (int)n, (int)scale_size, (int)bias_size, eps)
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
static int add_norm_dispatch(infiniRocmRuntime_t rt, const T *a, const T *b, const T *scale, const T *bias, T *y,
                             int64_t outer, int64_t n, int64_t scale_size, int64_t bias_size, float eps) {
    constexpr int VEC = Elem<T>::VEC;
    const bool al = is_aligned16(a) && is_aligned16(b) && is_aligned16(y) && is_aligned16(scale) &&
                    (bias == nullptr || is_aligned16(bias)) && (n % VEC == 0);
    const int chunks = (int)ceil_div(n, (int64_t)64 * VEC);
    if (!al || chunks > 4)
        return INFINI_ROCM_UNSUPPORTED;
    const unsigned g = pgrid(ceil_div(outer, 4), rt->num_cu);
#define AN(C)                                                                                                  \
    hipLaunchKernelGGL((add_norm_wave_kernel<T, C, RMS>), dim3(g), dim3(256), 0, rt->stream, a, b, scale, bias, y, \
                       (long)outer, (int)n, (int)scale_size, (int)bias_size, eps)
    if (chunks <= 1) AN(1);
    else if (chunks <= 2) AN(2);
    else AN(4);
#undef AN
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
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(outer >= 0 && norm_size >= 0 && norm_size < (1ll << 31), "add_norm: bad extent");
    if (outer == 0 || norm_size == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(a && b && y && scale, "add_norm: NULL tensor");
    IROCM_CHECK_ARG(scale_size == 1 || scale_size == norm_size, "add_norm: bad scale size");
    IROCM_CHECK_ARG(bias == nullptr || bias_size == 1 || bias_size == norm_size, "add_norm: bad bias size");
    int st = INFINI_ROCM_UNSUPPORTED;
#define GO(T)                                                                                                   \
    st = rms ? add_norm_dispatch<T, true>(rt, (const T *)a, (const T *)b, (const T *)scale, (const T *)bias, (T *)y, \
                                          outer, norm_size, scale_size, bias_size, eps)                        \
             : add_norm_dispatch<T, false>(rt, (const T *)a, (const T *)b, (const T *)scale, (const T *)bias, (T *)y, \
                                           outer, norm_size, scale_size, bias_size, eps)
    switch (dtype) {
    case INFINI_DT_F32: GO(float); break;
    case INFINI_DT_F16: GO(__half); break;
    case INFINI_DT_BF16: GO(__hip_bfloat16); break;
    default: IROCM_FAIL(INFINI_ROCM_UNSUPPORTED, "add_norm: unsupported dtype %s", dtype_name(dtype));
    }
#undef GO
    if (st != INFINI_ROCM_UNSUPPORTED)
        return st;
    // outside the fused kernel's reach (long or unaligned rows): the two-kernel chain, in place over y
    const int64_t shape[2] = {outer, norm_size}, str[2] = {norm_size, 1};
    st = infini_rocm_binary(rt, INFINI_BIN_ADD, dtype, a, b, y, 2, shape, str, str);
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
