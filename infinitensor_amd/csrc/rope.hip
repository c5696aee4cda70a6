// RoPE (rotate-half form) for gfx950.
// Replaces _rope_kernel (reference: src/kernels/cuda/rope.cu:6-31, glue rope.cc:8-33):
//   for column c of a head (c = j % dim_head, half = dim_head / 2), angle = pos * theta^(-2 (c % half) / dim_head):
//     c <  half : y[j] = x[j] cos - x[j + half] sin
//     c >= half : y[j] = x[j] cos + x[j - half] sin
// Deliberate deviations from the reference: the launch covers every (batch, position)
// (the reference grid covers one: rope.cu:85); dim_head and theta are arguments (reference hard-codes 128 /
// 10000: rope.cc:25); the partner element is never read outside its head (the reference test relies on an
// out-of-bounds read returning 0: test_cuda_rope.cc:17-31 uses dim_model 32 < dim_head 128 — a trailing partial
// head is accepted here and its missing partner columns count as 0, which reproduces that test without the read).
// Two kernels. rope_rows_kernel (whole heads, 16-byte-aligned rows — every decoder shape): a workgroup takes TOK tokens at
// a time; the angle of column pair c depends on (token, c) only, so its sine / cosine is computed ONCE per (token, c) —
// TOK * half values per step, one per thread — parked in LDS and shared by all heads of the token (round 3 computed a
// sincosf per (token, head, pair): 32 x the transcendental work at 32 heads, behind 2-byte loads and two 64-bit divisions per
// element: 1.5 TB/s); the row then moves in 16-byte vectors, the low and the high half of a head by the same thread.
// rope_kernel: the general form (a trailing partial head, odd alignments), one thread per (token, head, pair).
// fp32 math. HBM-bound: 2 * numel * sizeof(T) bytes.
#include "common.h"

namespace irocm {

template <typename T> struct RLd;
template <> struct RLd<float> {
    __device__ static inline float ld(const float *p) { return *p; }
    __device__ static inline void st(float *p, float v) { *p = v; }
};
template <> struct RLd<__half> {
    __device__ static inline float ld(const __half *p) { return __half2float(*p); }
    __device__ static inline void st(__half *p, float v) { *p = __float2half_rn(v); }
};
template <> struct RLd<__hip_bfloat16> {
    __device__ static inline float ld(const __hip_bfloat16 *p) { return __bfloat162float(*p); }
    __device__ static inline void st(__hip_bfloat16 *p, float v) { *p = __float2bfloat16(v); }
};

template <typename T, typename P>
__global__ __launch_bounds__(256) void rope_kernel(const P *__restrict__ pos, const T *__restrict__ x,
                                                   T *__restrict__ y, long tokens, int dim_model, int dim_head,
                                                   float neg2_log2theta_over_dh, int hs_seq) {
    const int half = dim_head / 2;
    const int heads = (dim_model + dim_head - 1) / dim_head; // a trailing partial head is allowed
    const long pairs_per_token = (long)heads * half;
    const long total = tokens * pairs_per_token;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long tok = i / pairs_per_token;
        const int pr = (int)(i - tok * pairs_per_token);
        const int head = pr / half, c = pr - head * half;
        const int j0 = head * dim_head + c, j1 = j0 + half;
        if (j0 >= dim_model)
            continue;
        const float ang = (float)pos[tok] * exp2f((float)c * neg2_log2theta_over_dh);
        float sn, cs;
        sincosf(ang, &sn, &cs);
        const long j = tok * dim_model + j0;
        const bool pair = j1 < dim_model; // partner column beyond the row: treated as 0, never read
        const float a = RLd<T>::ld(x + j), b = pair ? RLd<T>::ld(x + j + half) : 0.f;
        // hs_seq > 0: head-split store, token (b, s) head h column c -> y[b][h][s][c] (the Reshape([B, S, H, D]) ->
        // Transpose(0, 2, 1, 3) a decoder applies next); dim_model % dim_head == 0 then
        const long jo = hs_seq ? (((tok / hs_seq) * heads + head) * hs_seq + tok % hs_seq) * dim_head + c : j;
        RLd<T>::st(y + jo, a * cs - b * sn);
        if (pair)
            RLd<T>::st(y + jo + half, b * cs + a * sn);
    }
}

template <typename T, int N> struct alignas(sizeof(T) * N) RVec { T v[N]; };

// TOK tokens per workgroup step; dim_model % dim_head == 0, half % VEC == 0, rows 16-byte aligned.
template <typename T, typename P, int TOK>
__global__ __launch_bounds__(256) void rope_rows_kernel(const P *__restrict__ pos, const T *__restrict__ x, T *__restrict__ y,
                                                        int tokens, int dim_model, int dim_head, float neg2_log2theta_over_dh,
                                                        int hs_seq) {
    constexpr int VEC = 16 / (int)sizeof(T);
    using V = RVec<T, VEC>;
    extern __shared__ float sc[]; // [TOK][half] cos, then [TOK][half] sin
    const int half = dim_head / 2, heads = dim_model / dim_head;
    const int nv = half / VEC;            // vectors per half head
    const int pairs = heads * nv;         // (lo, hi) vector pairs per token
    float *cs_s = sc, *sn_s = sc + TOK * half;
    for (int t0 = blockIdx.x * TOK; t0 < tokens; t0 += gridDim.x * TOK) {
        for (int i = threadIdx.x; i < TOK * half; i += 256) {
            const int tl = i / half, c = i - tl * half;
            const int tok = t0 + tl;
            float sn = 0.f, cs = 1.f;
            if (tok < tokens) {
                const float ang = (float)pos[tok] * exp2f((float)c * neg2_log2theta_over_dh);
                sincosf(ang, &sn, &cs);
            }
            cs_s[i] = cs;
            sn_s[i] = sn;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < TOK * pairs; i += 256) {
            const int tl = i / pairs, pr = i - tl * pairs;
            const int tok = t0 + tl;
            if (tok >= tokens)
                break;
            const int head = pr / nv, v = pr - head * nv;
            const long j = (long)tok * dim_model + head * dim_head + v * VEC;
            const V lo = *reinterpret_cast<const V *>(x + j), hi = *reinterpret_cast<const V *>(x + j + half);
            V olo, ohi;
            const float *cp = cs_s + tl * half + v * VEC, *sp = sn_s + tl * half + v * VEC;
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const float a = RLd<T>::ld(&lo.v[q]), b = RLd<T>::ld(&hi.v[q]), cs = cp[q], sn = sp[q];
                RLd<T>::st(&olo.v[q], a * cs - b * sn);
                RLd<T>::st(&ohi.v[q], b * cs + a * sn);
            }
            long jo = j;
            if (hs_seq) { // token (b, s), head h -> y[b][h][s][:]
                const int bb = tok / hs_seq, ss = tok - bb * hs_seq;
                jo = (((long)bb * heads + head) * hs_seq + ss) * dim_head + v * VEC;
            }
            *reinterpret_cast<V *>(y + jo) = olo;
            *reinterpret_cast<V *>(y + jo + half) = ohi;
        }
        __syncthreads();
    }
}

} // namespace irocm

using namespace irocm;

extern "C" int infini_rocm_rope(infiniRocmRuntime_t rt, int dtype, int pos_dtype, const void *pos, const void *x,
                                void *y, int64_t tokens, int64_t dim_model, int64_t dim_head, float theta) {
    return infini_rocm_rope_headsplit(rt, dtype, pos_dtype, pos, x, y, tokens, dim_model, dim_head, theta, 0);
}

extern "C" int infini_rocm_rope_headsplit(infiniRocmRuntime_t rt, int dtype, int pos_dtype, const void *pos, const void *x,
                                          void *y, int64_t tokens, int64_t dim_model, int64_t dim_head, float theta,
                                          int64_t seq) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(seq >= 0 && seq < (1ll << 31), "rope: bad sequence length");
    IROCM_CHECK_ARG(seq == 0 || (tokens % seq == 0 && dim_model % dim_head == 0 && x != y),
                    "rope: the head-split store needs tokens %% seq == 0, whole heads and separate buffers");
    IROCM_CHECK_ARG(tokens >= 0 && dim_model > 0 && dim_head > 0, "rope: bad extent");
    IROCM_CHECK_ARG(dim_head % 2 == 0, "rope: head dim %lld must be even", (long long)dim_head);
    IROCM_CHECK_ARG(theta > 1.0f, "rope: theta must be > 1");
    if (tokens == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(pos && x && y, "rope: NULL tensor");
    const float k = -2.0f * log2f(theta) / (float)dim_head;
    {
        const size_t es = dtype_size(dtype);
        const int vec = es ? (int)(16 / es) : 1;
        const bool rows = es && dim_model % dim_head == 0 && (dim_head / 2) % vec == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0 &&
                          tokens < (1ll << 31) && dim_model < (1ll << 24) && dim_head <= 1024 &&
                          (dtype == INFINI_DT_F32 || dtype == INFINI_DT_F16 || dtype == INFINI_DT_BF16) &&
                          (pos_dtype == INFINI_DT_I32 || pos_dtype == INFINI_DT_U32 || pos_dtype == INFINI_DT_I64);
        if (rows) {
            constexpr int TOK = 4;
            const size_t lds = (size_t)2 * TOK * (dim_head / 2) * sizeof(float);
            long g = ceil_div(tokens, TOK);
            if (g > (long)rt->num_cu * 8) g = (long)rt->num_cu * 8;
#define GOR(T, P)                                                                                                    \
    hipLaunchKernelGGL((rope_rows_kernel<T, P, TOK>), dim3((unsigned)g), dim3(256), lds, rt->stream, (const P *)pos, \
                       (const T *)x, (T *)y, (int)tokens, (int)dim_model, (int)dim_head, k, (int)seq)
#define GORP(T)                                                                                                      \
    switch (pos_dtype) {                                                                                             \
    case INFINI_DT_I32: GOR(T, int32_t); break;                                                                      \
    case INFINI_DT_U32: GOR(T, uint32_t); break;                                                                     \
    default: GOR(T, int64_t); break;                                                                                 \
    }
            switch (dtype) {
            case INFINI_DT_F32: GORP(float); break;
            case INFINI_DT_F16: GORP(__half); break;
            default: GORP(__hip_bfloat16); break;
            }
#undef GORP
#undef GOR
            IROCM_LAUNCH_CHECK("rope_rows");
            return INFINI_ROCM_OK;
        }
    }
    const long total = tokens * (ceil_div(dim_model, dim_head) * (dim_head / 2));
    long g = ceil_div(total, 256);
    if (g > (long)rt->num_cu * 16) g = (long)rt->num_cu * 16;
#define GO(T, P)                                                                                   \
    hipLaunchKernelGGL((rope_kernel<T, P>), dim3((unsigned)g), dim3(256), 0, rt->stream,           \
                       (const P *)pos, (const T *)x, (T *)y, (long)tokens, (int)dim_model,         \
                       (int)dim_head, k, (int)seq)
#define GOP(T)                                                                                     \
    switch (pos_dtype) {                                                                           \
    case INFINI_DT_I32: GO(T, int32_t); break;                                                     \
    case INFINI_DT_U32: GO(T, uint32_t); break;                                                    \
    case INFINI_DT_I64: GO(T, int64_t); break;                                                     \
    default: IROCM_FAIL(INFINI_ROCM_UNSUPPORTED, "rope: positions must be int32/uint32/int64");    \
    }
    switch (dtype) {
    case INFINI_DT_F32: GOP(float); break;
    case INFINI_DT_F16: GOP(__half); break;
    case INFINI_DT_BF16: GOP(__hip_bfloat16); break;
    default: IROCM_FAIL(INFINI_ROCM_UNSUPPORTED, "rope: unsupported dtype %s", dtype_name(dtype));
    }
#undef GOP
#undef GO
    IROCM_LAUNCH_CHECK("rope");
    return INFINI_ROCM_OK;
}
