// Reduce (Mean/Sum), BatchNormalization (inference), MaxPool / AveragePool for gfx950.
//
// Replaces (reference): ReduceCudnnBase::compute src/kernels/cuda/reduce.cc:10-108 (cudnnReduceTensor),
// BatchNormCudnn src/kernels/cuda/batch_norm.cc:7-67 (cudnnBatchNormalizationForwardInference, SPATIAL),
// poolingCudnn src/kernels/cuda/pooling.cc:6-95 (cudnnPoolingForward; AVERAGE_COUNT_INCLUDE_PADDING).
// All HBM-bound: algorithmic bytes = (numel_in + numel_out) * sizeof(T). fp32 arithmetic throughout.
#include "common.h"
#include "gemm_common.h" // (udivmod_m / udiv_magic: division by multiply-high)
#include <type_traits>

namespace irocm {

constexpr int MD = INFINI_ROCM_MAX_DIMS;

template <typename T> struct LdSt;
template <> struct LdSt<float> {
    __device__ static inline float ld(const float *p) { return *p; }
    __device__ static inline void st(float *p, float v) { *p = v; }
};
template <> struct LdSt<__half> {
    __device__ static inline float ld(const __half *p) { return __half2float(*p); }
    __device__ static inline void st(__half *p, float v) { *p = __float2half_rn(v); }
};
template <> struct LdSt<__hip_bfloat16> {
    __device__ static inline float ld(const __hip_bfloat16 *p) { return __bfloat162float(*p); }
    __device__ static inline void st(__hip_bfloat16 *p, float v) { *p = __float2bfloat16(v); }
};

template <typename T, int N> struct alignas(sizeof(T) * N) PackN {
    T v[N];
};

__device__ inline float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o, 64);
    return v;
}

// ---- Reduce: trailing dims reduced -> [rows, n] row sums, one wave per row ---------------------
template <typename T>
__global__ __launch_bounds__(256) void reduce_rows_kernel(const T *__restrict__ x, T *__restrict__ y, long rows,
                                                          long n, float scale) {
    constexpr int VEC = 16 / (int)sizeof(T);
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows)
        return;
    const T *xr = x + row * n;
    float s = 0.f;
    const bool al = (((uintptr_t)xr) & 15) == 0;
    long i = 0;
    if (al) {
        const long nv = n / VEC;
        for (long v = lane; v < nv; v += 64) {
            PackN<T, VEC> pk = reinterpret_cast<const PackN<T, VEC> *>(xr)[v];
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                s += LdSt<T>::ld(&pk.v[j]);
        }
        i = nv * VEC;
    }
    for (long k = i + lane; k < n; k += 64)
        s += LdSt<T>::ld(xr + k);
    s = wsum(s);
    if (lane == 0)
        LdSt<T>::st(y + row, s * scale);
}

// Sum of the n 16-bit elements e0 .. e0 + n - 1 of a span staged in LDS (4-byte aligned base), fp32 accumulation: two elements per
// LDS read and ONE instruction per pair — v_dot2c_f32_{f16,bf16}(word, (1, 1), s) adds both halves of a word to the fp32 sum. The word a
// row shares with its neighbour (or, at the very end of an odd span, with bytes nobody wrote) has the FOREIGN half cleared in the DATA
// (+0.0), not in the weights: 0 x Inf / NaN is NaN, so a zero weight would let a neighbour's -inf mask or the stale LDS bits behind the
// span poison this row's sum.
template <typename T> __device__ __forceinline__ float row_sum16(const unsigned *words, int e0, int n) {
    constexpr unsigned kOne = std::is_same<T, __half>::value ? 0x3c00u : 0x3f80u;
    constexpr unsigned kOnes = kOne | (kOne << 16);
    auto dot = [kOnes](unsigned word, float acc) {
        if constexpr (std::is_same<T, __half>::value) {
            typedef _Float16 h2_ __attribute__((ext_vector_type(2)));
            return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_, word), __builtin_bit_cast(h2_, kOnes), acc, false);
        } else {
            typedef __bf16 b2_ __attribute__((ext_vector_type(2)));
            return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2_, word), __builtin_bit_cast(b2_, kOnes), acc, false);
        }
    };
    const int e1 = e0 + n - 1;
    const int w0 = e0 >> 1, w1 = e1 >> 1;
    const unsigned first = (e0 & 1) ? 0xffff0000u : 0xffffffffu; // a row starting on an odd element owns the high half only
    const unsigned last = (e1 & 1) ? 0xffffffffu : 0x0000ffffu;  // a row ending on an even element owns the low half only
    float s = 0.f;
    if (w0 == w1)
        return dot(words[w0] & first & last, s);
    s = dot(words[w0] & first, s);
    for (int w = w0 + 1; w < w1; ++w)
        s = dot(words[w], s);
    return dot(words[w1] & last, s);
}

// ---- Reduce: trailing dims reduced, SHORT rows (n < 64: the 7 x 7 planes of a global average pool written as ReduceMean) --
// A block takes RB consecutive rows = one contiguous span of RB * n elements: coalesced loads into LDS as fp32, then one
// thread per row sums its n values (the general kernel below read 49 strided elements per thread: 0.6 TB/s).
template <typename T, int RB>
__global__ __launch_bounds__(256) void reduce_short_rows_kernel(const T *__restrict__ x, T *__restrict__ y, long rows, int n,
                                                                float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rbuf_raw[]; // RB * n elements of T (kept in the storage type:
    T *rbuf = reinterpret_cast<T *>(rbuf_raw);                               //  25 KB per block at n = 49 f16 -> 6 blocks per CU)
    constexpr int VEC = 16 / (int)sizeof(T);
    const long nblk = (rows + RB - 1) / RB;
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long row0 = blk * RB;
        const int nrows = (int)(rows - row0 < RB ? rows - row0 : RB);
        const int cnt = nrows * n;
        const T *src = x + row0 * n;
        const bool al = (((uintptr_t)src) & 15) == 0;
        const int nv = al ? cnt / VEC : 0;
        for (int v = threadIdx.x; v < nv; v += 256)
            reinterpret_cast<PackN<T, VEC> *>(rbuf)[v] = reinterpret_cast<const PackN<T, VEC> *>(src)[v];
        for (int i = nv * VEC + threadIdx.x; i < cnt; i += 256)
            rbuf[i] = src[i];
        __syncthreads();
        if ((int)threadIdx.x < nrows) {
            float s = 0.f;
            if constexpr (sizeof(T) == 2) {
                // (one 2-byte read + conversion + add per element made the 7 x 7 planes of a global average pool instruction-bound:
                // 0.48 of the HBM peak)
                s = row_sum16<T>(reinterpret_cast<const unsigned *>(rbuf), (int)threadIdx.x * n, n);
            } else {
                const T *r = rbuf + threadIdx.x * n;
                for (int k = 0; k < n; ++k)
                    s += LdSt<T>::ld(r + k);
            }
            LdSt<T>::st(y + row0 + threadIdx.x, s * scale);
        }
        __syncthreads();
    }
}

// ---- Reduce: general axes. One thread per output element (coalesced when the innermost kept dim
// is the innermost input dim), serial loop over the reduced index space. -------------------------
struct ReduceArgs {
    int nk, nr;           // number of kept / reduced (collapsed) dims
    long kshape[MD], kstride[MD];
    long rshape[MD], rstride[MD];
    long nout, nred;
    float scale;
};

template <typename T>
__global__ __launch_bounds__(256) void reduce_general_kernel(const T *__restrict__ x, T *__restrict__ y,
                                                             ReduceArgs p) {
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < p.nout; o += (long)gridDim.x * 256) {
        long rem = o, base = 0;
        for (int d = p.nk - 1; d >= 0; --d) {
            const long q = rem / p.kshape[d];
            base += (rem - q * p.kshape[d]) * p.kstride[d];
            rem = q;
        }
        float s = 0.f;
        for (long r = 0; r < p.nred; ++r) {
            long rr = r, off = base;
            for (int d = p.nr - 1; d >= 0; --d) {
                const long q = rr / p.rshape[d];
                off += (rr - q * p.rshape[d]) * p.rstride[d];
                rr = q;
            }
            s += LdSt<T>::ld(x + off);
        }
        LdSt<T>::st(y + o, s * p.scale);
    }
}

template <typename T>
static int reduce_dispatch(infiniRocmRuntime_t rt, const void *x, void *y, int ndim, const int64_t *shape,
                           const int *reduced, bool mean) {
    long stride[MD], total = 1;
    for (int d = ndim - 1; d >= 0; --d) {
        stride[d] = total;
        total *= shape[d];
    }
    ReduceArgs p;
    p.nk = p.nr = 0;
    p.nout = p.nred = 1;
    bool trailing = true; // all reduced dims (extent > 1) come after all kept dims (extent > 1)
    bool seen_red = false;
    for (int d = 0; d < ndim; ++d) {
        if (shape[d] == 1)
            continue;
        if (reduced[d]) {
            seen_red = true;
            if (p.nr > 0 && p.rstride[p.nr - 1] == stride[d] * shape[d]) {
                p.rshape[p.nr - 1] *= shape[d];
                p.rstride[p.nr - 1] = stride[d];
            } else {
                p.rshape[p.nr] = shape[d];
                p.rstride[p.nr] = stride[d];
                ++p.nr;
            }
            p.nred *= shape[d];
        } else {
            if (seen_red)
                trailing = false;
            if (p.nk > 0 && p.kstride[p.nk - 1] == stride[d] * shape[d]) {
                p.kshape[p.nk - 1] *= shape[d];
                p.kstride[p.nk - 1] = stride[d];
            } else {
                p.kshape[p.nk] = shape[d];
                p.kstride[p.nk] = stride[d];
                ++p.nk;
            }
            p.nout *= shape[d];
        }
    }
    if (total == 0)
        return INFINI_ROCM_OK; // empty input: nothing to write that the reference defines
    p.scale = mean ? 1.0f / (float)p.nred : 1.0f;
    if (trailing && p.nred < 64 && p.nred > 1 && p.nout >= 256) {
        constexpr int RB = 256;
        long g = ceil_div(p.nout, RB);
        // persistent workgroups, as many as are RESIDENT at once: a CU holds 160 KB / (RB rows in LDS) of them (six at 49 f16
        // elements). With a fixed eight per CU the last two of every CU started when the first six had finished all their trips and
        // ran at a third of the occupancy: the HBM-sized ReduceMean over 7 x 7 planes sat at 0.54-0.57 of the HBM peak.
        const long per_cu = std::max<long>(1, std::min<long>(8, (160 * 1024) / ((long)RB * p.nred * (long)sizeof(T) + 512)));
        if (g > (long)rt->num_cu * per_cu) g = (long)rt->num_cu * per_cu;
        hipLaunchKernelGGL((reduce_short_rows_kernel<T, RB>), dim3((unsigned)g), dim3(256), (size_t)RB * p.nred * sizeof(T), rt->stream,
                           (const T *)x, (T *)y, p.nout, (int)p.nred, p.scale);
    } else if (trailing && p.nred >= 64) {
        hipLaunchKernelGGL((reduce_rows_kernel<T>), dim3((unsigned)ceil_div(p.nout, 4)), dim3(256), 0,
                           rt->stream, (const T *)x, (T *)y, p.nout, p.nred, p.scale);
    } else {
        long g = ceil_div(p.nout, 256);
        if (g > (long)rt->num_cu * 16) g = (long)rt->num_cu * 16;
        hipLaunchKernelGGL((reduce_general_kernel<T>), dim3((unsigned)g), dim3(256), 0, rt->stream,
                           (const T *)x, (T *)y, p);
    }
    IROCM_LAUNCH_CHECK("reduce");
    return INFINI_ROCM_OK;
}

// ---- BatchNorm inference: x [N, C, inner]; mean/var/scale/bias fp32 [C] ------------------------
template <typename T>
__global__ __launch_bounds__(256) void batch_norm_kernel(const T *__restrict__ x, const float *__restrict__ mean,
                                                         const float *__restrict__ var,
                                                         const float *__restrict__ scale,
                                                         const float *__restrict__ bias, T *__restrict__ y,
                                                         long nc, long c, long inner, float eps) {
    constexpr int VEC = 16 / (int)sizeof(T);
    // one block per (n, c) plane chunk: blockIdx.y = plane, blockIdx.x = chunk
    const long plane = blockIdx.y + (long)blockIdx.z * 65535;
    if (plane >= nc)
        return;
    const long ch = plane % c;
    const float a = scale[ch] * rsqrtf(var[ch] + eps);
    const float b = bias[ch] - mean[ch] * a;
    const T *xp = x + plane * inner;
    T *yp = y + plane * inner;
    const bool al = ((((uintptr_t)xp) | ((uintptr_t)yp)) & 15) == 0;
    if (al) {
        const long nv = inner / VEC;
        for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nv; v += (long)gridDim.x * 256) {
            PackN<T, VEC> pk = reinterpret_cast<const PackN<T, VEC> *>(xp)[v], o;
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                LdSt<T>::st(&o.v[j], LdSt<T>::ld(&pk.v[j]) * a + b);
            reinterpret_cast<PackN<T, VEC> *>(yp)[v] = o;
        }
        for (long i = nv * VEC + (long)blockIdx.x * 256 + threadIdx.x; i < inner; i += (long)gridDim.x * 256)
            LdSt<T>::st(yp + i, LdSt<T>::ld(xp + i) * a + b);
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < inner; i += (long)gridDim.x * 256)
            LdSt<T>::st(yp + i, LdSt<T>::ld(xp + i) * a + b);
    }
}

// ---- Pooling -----------------------------------------------------------------------------------
struct PoolArgs {
    long n, c, h, w, oh, ow;
    int kh, kw, dh, dw, ph, pw, sh, sw;
    int relu; // max pool of relu(x) == relu of the max pool: the preceding Relu folded in
};

// IDX = int when every index fits 31 bits (64-bit div/mod per output is most of the cost of this kernel otherwise)
template <typename T, bool MAX, typename IDX>
__global__ __launch_bounds__(256) void pool2d_kernel(const T *__restrict__ x, T *__restrict__ y, PoolArgs p) {
    const IDX total = (IDX)(p.n * p.c * p.oh * p.ow);
    const IDX ow = (IDX)p.ow, oh = (IDX)p.oh, w = (IDX)p.w, h = (IDX)p.h;
    const int sh = (int)p.sh, sw = (int)p.sw, ph = (int)p.ph, pw = (int)p.pw, dh = (int)p.dh, dw = (int)p.dw;
    for (IDX i = (IDX)blockIdx.x * 256 + threadIdx.x; i < total; i += (IDX)gridDim.x * 256) {
        const IDX q = i / ow, ox = i - q * ow;
        const IDX plane = q / oh, oy = q - plane * oh;
        const T *xp = x + (long)plane * h * w;
        float acc = MAX ? -INFINITY : 0.f;
        for (int r = 0; r < p.kh; ++r) {
            const IDX iy = oy * sh - ph + r * dh;
            if (iy < 0 || iy >= h)
                continue;
            for (int s = 0; s < p.kw; ++s) {
                const IDX ix = ox * sw - pw + s * dw;
                if (ix < 0 || ix >= w)
                    continue;
                const float v = LdSt<T>::ld(xp + iy * w + ix);
                acc = MAX ? fmaxf(acc, v) : acc + v;
            }
        }
        // AveragePool counts padding (reference pooling.cc:86-90: COUNT_INCLUDE_PADDING)
        if (MAX && p.relu)
            acc = fmaxf(acc, 0.f);
        LdSt<T>::st(y + i, MAX ? acc : acc / (float)(p.kh * p.kw));
    }
}

// MaxPool 3x3 / stride 2 / pad 1 / dilation 1 on 16-bit types, W % 8 == 0 (the ResNet stem pool). One thread per 4 outputs of TWO
// output rows: the five input rows 2 oy - 1 .. 2 oy + 3 as one 16-byte load each (8 columns) plus the left neighbour, horizontal
// maxima per input row first (row 2 oy + 1 serves both output rows), two 8-byte stores. Everything is branch-free — rows outside
// the image are loaded from a clamped row and replaced by -inf with selects — so the ten loads of a thread are in flight together.
// History: the generic kernel issues 9 two-byte loads per output (1.4 TB/s on this layer); round 2's form — one output row per
// thread, four integer divisions by run-time values per thread, ~150 instructions per 4 outputs — was VALU-bound at the HBM-sized
// shape (0.58 of 8 TB/s: 19 M threads x 150 instructions = 140 us of vector issue alone); this one spends ~140 per 8 outputs.
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const T *__restrict__ x, T *__restrict__ y, int planes, int h,
                                                           int w, int oh, int ow, int relu, unsigned quads_m, unsigned oh2_m) {
    const int quads = ow / 4, oh2 = (oh + 1) / 2;
    const int total = planes * oh2 * quads;
    struct alignas(16) V8 { T v[8]; };
    struct alignas(8) V4 { T v[4]; };
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        unsigned q, jq, pl, o2;
        udivmod_m((unsigned)i, (unsigned)quads, quads_m, q, jq);
        udivmod_m(q, (unsigned)oh2, oh2_m, pl, o2);
        const int oy = 2 * (int)o2, iy0 = 2 * oy - 1;
        const T *plane = x + (long)pl * h * w + 8 * jq;
        const int lback = jq > 0 ? 1 : 0;
        V8 c[5];
        T lf[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const int iy = min(max(iy0 + r, 0), h - 1);
            const T *row = plane + (long)iy * w;
            c[r] = *reinterpret_cast<const V8 *>(row);
            lf[r] = *(row - lback);
        }
        float hm[5][4];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const bool ok = (unsigned)(iy0 + r) < (unsigned)h;
            float f[9];
            f[0] = lback ? LdSt<T>::ld(&lf[r]) : -INFINITY;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                f[e + 1] = LdSt<T>::ld(&c[r].v[e]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float m = fmaxf(f[2 * k], fmaxf(f[2 * k + 1], f[2 * k + 2]));
                hm[r][k] = ok ? m : -INFINITY;
            }
        }
        const float m0 = relu ? 0.f : -INFINITY;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (oy + half >= oh)
                break;
            V4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                LdSt<T>::st(&o.v[k], fmaxf(fmaxf(m0, hm[2 * half][k]), fmaxf(hm[2 * half + 1][k], hm[2 * half + 2][k])));
            *reinterpret_cast<V4 *>(y + ((long)pl * oh + oy + half) * ow + 4 * jq) = o;
        }
    }
}

// Global average pool fast path (kernel == whole plane, no pad): one wave per plane.
template <typename T>
__global__ __launch_bounds__(256) void global_avgpool_kernel(const T *__restrict__ x, T *__restrict__ y,
                                                             long planes, long hw) {
    const int lane = threadIdx.x & 63;
    const long plane = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes)
        return;
    float s = 0.f;
    for (long i = lane; i < hw; i += 64)
        s += LdSt<T>::ld(x + plane * hw + i);
    s = wsum(s);
    if (lane == 0)
        LdSt<T>::st(y + plane, s / (float)hw);
}

// Global average pool over SMALL planes (ResNet's 7 x 7 head: 49 elements = 98 bytes per plane): a wave per plane moves one
// 2-byte load per lane and ran at 0.8 TB/s (26 MB in 32 us). Here a workgroup owns 256 consecutive planes = one contiguous
// slab: coalesced 16-byte loads into LDS, then one thread per plane sums its hw elements from LDS in index order (fp32).
typedef unsigned int pool_u32x4_t __attribute__((ext_vector_type(4)));
template <typename T>
__global__ __launch_bounds__(256) void global_avgpool_small_kernel(const T *__restrict__ x, T *__restrict__ y, long planes, int hw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long p0 = (long)blockIdx.x * 256;
    const long np = planes - p0 < 256 ? planes - p0 : 256;
    const long bytes = np * hw * (long)sizeof(T); // the slab starts 16-byte aligned: 256 * hw * sizeof(T) per workgroup
    const char *src = (const char *)(x + p0 * hw);
    for (long o = (long)threadIdx.x * 16; o < bytes; o += 256 * 16) {
        if (o + 16 <= bytes) {
            *(pool_u32x4_t *)(smem + o) = *(const pool_u32x4_t *)(src + o);
        } else {
            for (long b = o; b < bytes; b += sizeof(T))
                *(T *)(smem + b) = *(const T *)(src + b);
        }
    }
    __syncthreads();
    if ((long)threadIdx.x < np) {
        float s = 0.f;
        if constexpr (sizeof(T) == 2) {
            s = row_sum16<T>((const unsigned *)smem, (int)threadIdx.x * hw, hw);
        } else {
            const T *pl = (const T *)smem + (long)threadIdx.x * hw;
            for (int i = 0; i < hw; ++i)
                s += LdSt<T>::ld(pl + i);
        }
        LdSt<T>::st(y + p0 + threadIdx.x, s / (float)hw);
    }
}

// ---- LRN across channels (ONNX LRN; reference operator: src/operators/lrn.cc, only a Cambricon kernel exists,
// src/kernels/bang/lrn.cc:6-56): y[n,c,p] = x[n,c,p] / (bias + alpha / size * sum_{i in window(c)} x[n,i,p]^2)^beta,
// window(c) = [c - floor((size-1)/2), c + ceil((size-1)/2)] clipped to [0, C). One thread per (n, 4 consecutive pixels)
// walks the channels with the window's squares in a running fp32 sum that is RE-SUMMED (not slid) every step: exact and
// order-stable; the re-reads of the window hit L1 / L2 (neighbouring threads share the lines). HBM-bound.
template <typename T>
__global__ __launch_bounds__(256) void lrn_kernel(const T *__restrict__ x, T *__restrict__ y, long n, int c, long inner, int size,
                                                  float alpha_over_size, float beta, float bias) {
    const int lo = (size - 1) / 2, hi = size - 1 - lo;
    const long total = n * inner;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long img = i / inner, pix = i - img * inner;
        const T *xp = x + img * c * inner + pix;
        T *yp = y + img * c * inner + pix;
        for (int ch = 0; ch < c; ++ch) {
            const int a = ch - lo < 0 ? 0 : ch - lo, b = ch + hi >= c ? c - 1 : ch + hi;
            float sq = 0.f;
            for (int k = a; k <= b; ++k) {
                const float v = LdSt<T>::ld(xp + (long)k * inner);
                sq = fmaf(v, v, sq);
            }
            const float d = fmaf(alpha_over_size, sq, bias);
            // d^-beta = exp2(-beta * log2 d); d > 0 for every legal attribute set (bias > 0 or data != 0)
            LdSt<T>::st(yp + (long)ch * inner, LdSt<T>::ld(xp + (long)ch * inner) * exp2f(-beta * log2f(d)));
        }
    }
}

} // namespace irocm

using namespace irocm;

extern "C" {

int infini_rocm_reduce(infiniRocmRuntime_t rt, int kind, int dtype, const void *x, void *y, int ndim,
                       const int64_t *shape, const int *reduced) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(kind == 0 || kind == 1, "reduce: kind must be 0 (sum) or 1 (mean)");
    IROCM_CHECK_ARG(ndim >= 0 && ndim <= MD, "reduce: rank %d out of range", ndim);
    for (int d = 0; d < ndim; ++d)
        IROCM_CHECK_ARG(shape[d] >= 0, "reduce: negative extent");
    IROCM_CHECK_ARG(x && y, "reduce: NULL tensor");
    switch (dtype) {
    case INFINI_DT_F32: return reduce_dispatch<float>(rt, x, y, ndim, shape, reduced, kind == 1);
    case INFINI_DT_F16: return reduce_dispatch<__half>(rt, x, y, ndim, shape, reduced, kind == 1);
    case INFINI_DT_BF16: return reduce_dispatch<__hip_bfloat16>(rt, x, y, ndim, shape, reduced, kind == 1);
    default: IROCM_FAIL(INFINI_ROCM_UNSUPPORTED, "reduce: unsupported dtype %s", dtype_name(dtype));
    }
}

int infini_rocm_batch_norm(infiniRocmRuntime_t rt, int dtype, const void *x, const void *mean,
                           const void *var, const void *scale, const void *bias, void *y, int64_t n,
                           int64_t c, int64_t inner, float eps) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(n >= 0 && c >= 0 && inner >= 0, "batch_norm: negative extent");
    if (n * c * inner == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(x && y && mean && var && scale && bias, "batch_norm: NULL tensor");
    const long nc = n * c;
    const int vec = 16 / (int)dtype_size(dtype);
    long gx = ceil_div(ceil_div(inner, vec), 256);
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    dim3 grid((unsigned)gx, (unsigned)(nc < 65535 ? nc : 65535), (unsigned)ceil_div(nc, 65535));
#define GO(T)                                                                                      \
    hipLaunchKernelGGL((batch_norm_kernel<T>), grid, dim3(256), 0, rt->stream, (const T *)x,       \
                       (const float *)mean, (const float *)var, (const float *)scale,              \
                       (const float *)bias, (T *)y, nc, (long)c, (long)inner, eps)
    switch (dtype) {
    case INFINI_DT_F32: GO(float); break;
    case INFINI_DT_F16: GO(__half); break;
    case INFINI_DT_BF16: GO(__hip_bfloat16); break;
    default: IROCM_FAIL(INFINI_ROCM_UNSUPPORTED, "batch_norm: unsupported dtype %s", dtype_name(dtype));
    }
#undef GO
    IROCM_LAUNCH_CHECK("batch_norm");
    return INFINI_ROCM_OK;
}

int infini_rocm_lrn(infiniRocmRuntime_t rt, int dtype, const void *x, void *y, int64_t n, int64_t c, int64_t inner,
                    int size, float alpha, float beta, float bias) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(n >= 0 && c > 0 && inner >= 0 && c < (1ll << 31), "lrn: bad extent");
    IROCM_CHECK_ARG(size >= 1, "lrn: size must be >= 1");
    if (n == 0 || inner == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(x && y, "lrn: NULL tensor");
    long g = ceil_div(n * inner, 256);
    if (g > (long)rt->num_cu * 32) g = (long)rt->num_cu * 32;
#define GO(T)                                                                                      \
    hipLaunchKernelGGL((lrn_kernel<T>), dim3((unsigned)g), dim3(256), 0, rt->stream, (const T *)x, (T *)y, (long)n, (int)c,   \
                       (long)inner, size, alpha / (float)size, beta, bias)
    switch (dtype) {
    case INFINI_DT_F32: GO(float); break;
    case INFINI_DT_F16: GO(__half); break;
    case INFINI_DT_BF16: GO(__hip_bfloat16); break;
    default: IROCM_FAIL(INFINI_ROCM_UNSUPPORTED, "lrn: unsupported dtype %s", dtype_name(dtype));
    }
#undef GO
    IROCM_LAUNCH_CHECK("lrn");
    return INFINI_ROCM_OK;
}

int infini_rocm_pool2d(infiniRocmRuntime_t rt, int kind, int dtype, const void *x, void *y, int64_t n,
                       int64_t c, int64_t h, int64_t w, int kh, int kw, int dh, int dw, int ph, int pw,
                       int sh, int sw, int ceil_mode) {
    return infini_rocm_pool2d_relu(rt, kind, dtype, x, y, n, c, h, w, kh, kw, dh, dw, ph, pw, sh, sw, ceil_mode, 0);
}

int infini_rocm_pool2d_relu(infiniRocmRuntime_t rt, int kind, int dtype, const void *x, void *y, int64_t n,
                            int64_t c, int64_t h, int64_t w, int kh, int kw, int dh, int dw, int ph, int pw,
                            int sh, int sw, int ceil_mode, int relu) {
    IROCM_CHECK_ARG(!relu || kind == 0, "pool2d: the fused Relu needs max pooling");
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(kind == 0 || kind == 1, "pool2d: kind must be 0 (max) or 1 (average)");
    IROCM_CHECK_ARG(n >= 0 && c >= 0 && h >= 0 && w >= 0, "pool2d: negative extent");
    IROCM_CHECK_ARG(kh > 0 && kw > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0 && ph >= 0 && pw >= 0,
                    "pool2d: bad window attributes");
    // output size: reference src/operators/pooling.cc:17-35
    auto osz = [&](int64_t i, int k, int d, int p, int s) -> int64_t {
        const double v = ((double)(i + 2 * p - d * (k - 1) - 1)) / s + 1;
        return (int64_t)(ceil_mode ? ceil(v) : floor(v));
    };
    PoolArgs p;
    p.n = n; p.c = c; p.h = h; p.w = w;
    p.oh = osz(h, kh, dh, ph, sh);
    p.ow = osz(w, kw, dw, pw, sw);
    p.kh = kh; p.kw = kw; p.dh = dh; p.dw = dw; p.ph = ph; p.pw = pw; p.sh = sh; p.sw = sw;
    p.relu = relu;
    const long total = n * c * p.oh * p.ow;
    if (total <= 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(x && y, "pool2d: NULL tensor");
    const bool global_avg = kind == 1 && p.oh == 1 && p.ow == 1 && kh == h && kw == w && ph == 0 && pw == 0 &&
                            dh == 1 && dw == 1;
    long g = ceil_div(total, 256);
    if (g > (long)rt->num_cu * 16) g = (long)rt->num_cu * 16;
    const bool small = total + g * 256 < (1l << 31) && h * w < (1l << 31); // 32-bit index math is safe
    // specialised 3x3 / 2 max pool (16-bit types)
    const bool mp3 = kind == 0 && kh == 3 && kw == 3 && sh == 2 && sw == 2 && ph == 1 && pw == 1 && dh == 1 && dw == 1 &&
                     w % 8 == 0 && p.ow == w / 2 && (dtype == INFINI_DT_F16 || dtype == INFINI_DT_BF16) &&
                     (((uintptr_t)x) & 15) == 0 && (((uintptr_t)y) & 7) == 0 && total / 4 + 256l * rt->num_cu * 16 < (1l << 31);
    if (mp3) {
        const long items = n * c * ((p.oh + 1) / 2) * (p.ow / 4);
        long g3 = ceil_div(items, 256);
        if (g3 > (long)rt->num_cu * 16) g3 = (long)rt->num_cu * 16;
        const unsigned quads_m = udiv_magic((unsigned long long)(p.ow / 4)), oh2_m = udiv_magic((unsigned long long)((p.oh + 1) / 2));
        if (dtype == INFINI_DT_F16)
            hipLaunchKernelGGL(maxpool3x3s2_kernel<__half>, dim3((unsigned)g3), dim3(256), 0, rt->stream, (const __half *)x,
                               (__half *)y, (int)(n * c), (int)h, (int)w, (int)p.oh, (int)p.ow, relu, quads_m, oh2_m);
        else
            hipLaunchKernelGGL(maxpool3x3s2_kernel<__hip_bfloat16>, dim3((unsigned)g3), dim3(256), 0, rt->stream,
                               (const __hip_bfloat16 *)x, (__hip_bfloat16 *)y, (int)(n * c), (int)h, (int)w, (int)p.oh, (int)p.ow, relu, quads_m, oh2_m);
        IROCM_LAUNCH_CHECK("maxpool3x3s2");
        return INFINI_ROCM_OK;
    }
#define GO(T)                                                                                      \
    if (global_avg && (long)h * w * (long)sizeof(T) * 256 <= 64 * 1024 && ((((uintptr_t)x) & 15) == 0))  \
        hipLaunchKernelGGL((global_avgpool_small_kernel<T>), dim3((unsigned)ceil_div(n * c, 256)), \
                           dim3(256), (size_t)(h * w) * sizeof(T) * 256, rt->stream, (const T *)x, \
                           (T *)y, (long)(n * c), (int)(h * w));                                   \
    else if (global_avg)                                                                           \
        hipLaunchKernelGGL((global_avgpool_kernel<T>), dim3((unsigned)ceil_div(n * c, 4)),         \
                           dim3(256), 0, rt->stream, (const T *)x, (T *)y, (long)(n * c),          \
                           (long)(h * w));                                                         \
    else if (kind == 0 && small)                                                                   \
        hipLaunchKernelGGL((pool2d_kernel<T, true, int>), dim3((unsigned)g), dim3(256), 0, rt->stream, \
                           (const T *)x, (T *)y, p);                                               \
    else if (kind == 0)                                                                            \
        hipLaunchKernelGGL((pool2d_kernel<T, true, long>), dim3((unsigned)g), dim3(256), 0, rt->stream, \
                           (const T *)x, (T *)y, p);                                               \
    else if (small)                                                                                \
        hipLaunchKernelGGL((pool2d_kernel<T, false, int>), dim3((unsigned)g), dim3(256), 0, rt->stream, \
                           (const T *)x, (T *)y, p);                                               \
    else                                                                                           \
        hipLaunchKernelGGL((pool2d_kernel<T, false, long>), dim3((unsigned)g), dim3(256), 0, rt->stream, \
                           (const T *)x, (T *)y, p)
    switch (dtype) {
    case INFINI_DT_F32: GO(float); break;
    case INFINI_DT_F16: GO(__half); break;
    case INFINI_DT_BF16: GO(__hip_bfloat16); break;
    default: IROCM_FAIL(INFINI_ROCM_UNSUPPORTED, "pool2d: unsupported dtype %s", dtype_name(dtype));
    }
#undef GO
    IROCM_LAUNCH_CHECK("pool2d");
    return INFINI_ROCM_OK;
}

} // extern "C"
