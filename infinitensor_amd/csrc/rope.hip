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
// One thread per (token, head, pair): one angle, one sincos, two outputs. fp32 math. HBM-bound:
// 2 * numel * sizeof(T) bytes.
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
