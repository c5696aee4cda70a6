// AttentionKVCache: one decode step (query length 1) with in-place append to the K / V caches.
// Replaces _attention_kvcache_kernel_128_1/_2 (reference: src/kernels/cuda/attention_kvcache.cu:8-169, glue
// attention_kvcache.cc:8-60; operator src/operators/attention_kvcache.cc:5-35):
//   n = position_id[0] + 1;  cache[b, h, n-1, :] = k / v (appended in place);  out[b, h, 0, :] = softmax(q . K[0:n]^T / sqrt(D)) V[0:n]
// caches: [B, H, max_seq, D]; q, k, v, out: [B, H, 1, D]; position_id: device integer, element 0 is used for every
// (batch, head) exactly like the reference (attention_kvcache.cu:18).
// Deviations: the reference exponentiates raw scores (no max subtraction: overflows for |score| > 88)
// and merges 16-key chunks by their sums — the same value in exact arithmetic; here the usual running-max recurrence.
// The reference is fp32-only with D == 128; here f32 / f16 / bf16 storage (fp32 math), D a multiple of 128 up to 512.
//
// One workgroup (4 waves) per (batch, head). A 16-lane group owns one key at a time (16 keys in flight per workgroup):
// each lane holds D/16 consecutive elements of q, dots them with the key row (one coalesced D*sizeof(T) row per group),
// reduces over the 16 lanes with 4 xor-shuffles, and accumulates its D/16 outputs; the 16 group states are merged
// through LDS. HBM-bound: 2 * n * D * sizeof(T) bytes per (batch, head).
#include "common.h"

namespace irocm {

template <typename T> struct KvLd;
template <> struct KvLd<float> {
    __device__ static inline float ld(const float *p) { return *p; }
    __device__ static inline void st(float *p, float v) { *p = v; }
};
template <> struct KvLd<__half> {
    __device__ static inline float ld(const __half *p) { return __half2float(*p); }
    __device__ static inline void st(__half *p, float v) { *p = __float2half_rn(v); }
};
template <> struct KvLd<__hip_bfloat16> {
    __device__ static inline float ld(const __hip_bfloat16 *p) { return __bfloat162float(*p); }
    __device__ static inline void st(__hip_bfloat16 *p, float v) { *p = __float2bfloat16(v); }
};

template <typename T, typename P, int EPL> // EPL = D / 16 elements per lane
__global__ __launch_bounds__(256) void attention_kvcache_kernel(T *__restrict__ kc, T *__restrict__ vc,
                                                                const T *__restrict__ q, const T *__restrict__ kn,
                                                                const T *__restrict__ vn, const P *__restrict__ pos,
                                                                T *__restrict__ out, int max_seq) {
    constexpr int D = EPL * 16;
    __shared__ float s_m[16], s_l[16], s_o[16][D];
    const int bh = blockIdx.x;
    const int t = threadIdx.x, grp = t >> 4, sub = t & 15;
    const int n = (int)pos[0] + 1; // keys 0 .. n-1; key n-1 is the new one
    if (n < 1 || n > max_seq)
        return; // position outside the cache: nothing sensible to do (reference would write out of bounds)
    T *kcache = kc + (long)bh * max_seq * D, *vcache = vc + (long)bh * max_seq * D;
    const int e0 = sub * EPL;
    float qv[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e)
        qv[e] = KvLd<T>::ld(q + (long)bh * D + e0 + e);
    // append (every group's lanes of the first group write their slice once)
    if (grp == 0) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            kcache[(long)(n - 1) * D + e0 + e] = kn[(long)bh * D + e0 + e];
            vcache[(long)(n - 1) * D + e0 + e] = vn[(long)bh * D + e0 + e];
        }
    }
    const float scale = 1.0f / sqrtf((float)D);
    float m = -INFINITY, l = 0.f, o[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e)
        o[e] = 0.f;
    for (int key = grp; key < n; key += 16) {
        const bool fresh = key == n - 1; // read the new row from k / v: the append above may not be visible yet
        const T *kr = fresh ? kn + (long)bh * D : kcache + (long)key * D;
        const T *vr = fresh ? vn + (long)bh * D : vcache + (long)key * D;
        float dot = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e)
            dot = fmaf(qv[e], KvLd<T>::ld(kr + e0 + e), dot);
        dot += __shfl_xor(dot, 1);
        dot += __shfl_xor(dot, 2);
        dot += __shfl_xor(dot, 4);
        dot += __shfl_xor(dot, 8);
        const float sc = dot * scale;
        const float m_new = fmaxf(m, sc);
        const float alpha = expf(m - m_new), pv = expf(sc - m_new);
        l = l * alpha + pv;
#pragma unroll
        for (int e = 0; e < EPL; ++e)
            o[e] = o[e] * alpha + pv * KvLd<T>::ld(vr + e0 + e);
        m = m_new;
    }
    if (sub == 0) {
        s_m[grp] = m;
        s_l[grp] = l;
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e)
        s_o[grp][e0 + e] = o[e];
    __syncthreads();
    if (t < D) {
        float mm = -INFINITY;
#pragma unroll
        for (int g = 0; g < 16; ++g)
            mm = fmaxf(mm, s_m[g]);
        float ll = 0.f, oo = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const float wgt = s_m[g] == -INFINITY ? 0.f : expf(s_m[g] - mm);
            ll += s_l[g] * wgt;
            oo += s_o[g][t] * wgt;
        }
        KvLd<T>::st(out + (long)bh * D + t, oo / ll);
    }
}

} // namespace irocm

using namespace irocm;

extern "C" int infini_rocm_attention_kvcache(infiniRocmRuntime_t rt, int dtype, void *k_cache, void *v_cache,
                                             const void *q, const void *k, const void *v, int pos_dtype,
                                             const void *position_id, void *out, int64_t batch_heads, int64_t max_seq,
                                             int64_t head_dim) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(batch_heads >= 0 && max_seq > 0 && batch_heads < (1ll << 31) && max_seq < (1ll << 31),
                    "attention_kvcache: bad extent");
    IROCM_CHECK_ARG(head_dim == 128 || head_dim == 256, "attention_kvcache: head dim %lld not in {128, 256} (reference: 128 only)",
                    (long long)head_dim);
    if (batch_heads == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(k_cache && v_cache && q && k && v && position_id && out, "attention_kvcache: NULL tensor");
#define GO(T, P, E)                                                                                        \
    hipLaunchKernelGGL((attention_kvcache_kernel<T, P, E>), dim3((unsigned)batch_heads), dim3(256), 0,      \
                       rt->stream, (T *)k_cache, (T *)v_cache, (const T *)q, (const T *)k, (const T *)v,    \
                       (const P *)position_id, (T *)out, (int)max_seq)
#define GOP(T, E)                                                                                          \
    switch (pos_dtype) {                                                                                   \
    case INFINI_DT_I32: GO(T, int32_t, E); break;                                                          \
    case INFINI_DT_U32: GO(T, uint32_t, E); break;                                                         \
    case INFINI_DT_I64: GO(T, int64_t, E); break;                                                          \
    default: IROCM_FAIL(INFINI_ROCM_UNSUPPORTED, "attention_kvcache: position_id must be int32/uint32/int64"); \
    }
#define GOD(T)                                                                                             \
    if (head_dim == 128) { GOP(T, 8) } else { GOP(T, 16) }
    switch (dtype) {
    case INFINI_DT_F32: GOD(float); break;
    case INFINI_DT_F16: GOD(__half); break;
    case INFINI_DT_BF16: GOD(__hip_bfloat16); break;
    default: IROCM_FAIL(INFINI_ROCM_UNSUPPORTED, "attention_kvcache: unsupported dtype %s", dtype_name(dtype));
    }
#undef GOD
#undef GOP
#undef GO
    IROCM_LAUNCH_CHECK("attention_kvcache");
    return INFINI_ROCM_OK;
}
