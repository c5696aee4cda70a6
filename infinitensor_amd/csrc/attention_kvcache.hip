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
#include <algorithm>

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


// ---- round 5: the cache split over workgroups (reference: attention_kvcache.cu:18-25 splits it over gridDim.y = ceil(S / 16) and
// merges in a second kernel, :118-166) --------------------------------------------------------------------------------------------
// One workgroup per (batch, head) leaves a batch-1 decode step of a 32-head model on 32 of 256 CUs, and the element loads above
// keep 4 KB in flight per workgroup. Here the keys 0 .. n - 1 are cut into G equal chunks (n is read on the device, so the chunk
// length is computed in the kernel), workgroup (bh, g) reduces chunk g to a partial (m, l, o[D]) in fp32 — a 16-lane group takes
// KPI keys per iteration, every row as 16-byte vector loads issued back to back before the arithmetic (KPI x 16 keys = 8-16 KB of K
// and as much of V in flight per workgroup, several workgroups per CU) — and a second, tiny kernel merges the G partials per
// (batch, head). G = 1 (enough batch x heads to fill the chip by themselves) writes the output directly, one launch.
// A lane's EPL consecutive elements of one row, fetched as 16-byte vectors and kept PACKED until they are used (the next
// iteration's rows wait in registers under the current iteration's arithmetic: 4 registers per f16 row piece instead of 8).
template <typename T, int EPL> struct KvRow;
template <int EPL> struct KvRow<float, EPL> {
    struct Raw { float4 v[EPL / 4]; };
    __device__ static inline Raw ldraw(const float *p) {
        Raw r;
#pragma unroll
        for (int i = 0; i < EPL / 4; ++i)
            r.v[i] = ((const float4 *)p)[i];
        return r;
    }
    __device__ static inline void cvt(const Raw &r, float (&f)[EPL]) {
#pragma unroll
        for (int i = 0; i < EPL / 4; ++i) {
            f[4 * i] = r.v[i].x; f[4 * i + 1] = r.v[i].y; f[4 * i + 2] = r.v[i].z; f[4 * i + 3] = r.v[i].w;
        }
    }
};
template <int EPL> struct KvRow<__half, EPL> {
    struct Raw { uint4 v[EPL / 8]; };
    __device__ static inline Raw ldraw(const __half *p) {
        Raw r;
#pragma unroll
        for (int i = 0; i < EPL / 8; ++i)
            r.v[i] = ((const uint4 *)p)[i];
        return r;
    }
    __device__ static inline void cvt(const Raw &r, float (&f)[EPL]) {
#pragma unroll
        for (int i = 0; i < EPL / 8; ++i) {
            const unsigned u[4] = {r.v[i].x, r.v[i].y, r.v[i].z, r.v[i].w};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                f[8 * i + 2 * d] = (float)__builtin_bit_cast(_Float16, (unsigned short)(u[d] & 0xffffu));
                f[8 * i + 2 * d + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(u[d] >> 16));
            }
        }
    }
};
template <int EPL> struct KvRow<__hip_bfloat16, EPL> {
    struct Raw { uint4 v[EPL / 8]; };
    __device__ static inline Raw ldraw(const __hip_bfloat16 *p) {
        Raw r;
#pragma unroll
        for (int i = 0; i < EPL / 8; ++i)
            r.v[i] = ((const uint4 *)p)[i];
        return r;
    }
    __device__ static inline void cvt(const Raw &r, float (&f)[EPL]) {
#pragma unroll
        for (int i = 0; i < EPL / 8; ++i) {
            const unsigned u[4] = {r.v[i].x, r.v[i].y, r.v[i].z, r.v[i].w};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                f[8 * i + 2 * d] = __builtin_bit_cast(float, u[d] << 16);
                f[8 * i + 2 * d + 1] = __builtin_bit_cast(float, u[d] & 0xffff0000u);
            }
        }
    }
};

// part: [bh][G][D + 2] floats = o[D] (relative to the chunk's own maximum m), m, l
template <typename T, typename P, int EPL, int KPI, int NG> // KPI = keys per 16-lane group and iteration, NG = 16-lane groups per workgroup
__global__ __launch_bounds__(NG * 16) void attention_kvcache_split_kernel(T *__restrict__ kc, T *__restrict__ vc, const T *__restrict__ q,
                                                                      const T *__restrict__ kn, const T *__restrict__ vn,
                                                                      const P *__restrict__ pos, T *__restrict__ out,
                                                                      float *__restrict__ part, int max_seq, int G) {
    constexpr int D = EPL * 16;
    __shared__ float s_m[NG], s_l[NG], s_o[NG][D];
    const int bh = blockIdx.x, g = blockIdx.y;
    const int t = threadIdx.x, grp = t >> 4, sub = t & 15;
    const int n = (int)pos[0] + 1; // keys 0 .. n-1; key n-1 is the new one
    if (n < 1 || n > max_seq)
        return;
    T *kcache = kc + (long)bh * max_seq * D, *vcache = vc + (long)bh * max_seq * D;
    const int e0 = sub * EPL;
    float qv[EPL];
    KvRow<T, EPL>::cvt(KvRow<T, EPL>::ldraw(q + (long)bh * D + e0), qv);
    const float scale = 1.0f / sqrtf((float)D);
#pragma unroll
    for (int e = 0; e < EPL; ++e)
        qv[e] *= scale;
    // chunk g: keys [c0, c1); the chunk length is a multiple of the NG KPI keys a workgroup takes per iteration
    const int step = NG * KPI;
    const int len = ((n + G - 1) / G + step - 1) / step * step;
    const int c0 = g * len, c1 = min(n, c0 + len);
    // append: the workgroup whose chunk holds key n - 1 (the scores below read the new row from k / v, never from the cache)
    if (grp == 0 && n - 1 >= c0 && n - 1 < c1) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            kcache[(long)(n - 1) * D + e0 + e] = kn[(long)bh * D + e0 + e];
            vcache[(long)(n - 1) * D + e0 + e] = vn[(long)bh * D + e0 + e];
        }
    }
    float m = -INFINITY, l = 0.f, o[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e)
        o[e] = 0.f;
    // (A second register stage — the rows of iteration i + 1 requested before the arithmetic of iteration i — was built and measured:
    // 18.9 vs 18.6 us at B x H = 32, 4096 keys, and 126 instead of 89 registers. One workgroup per CU streams ~10 B/clk/CU, the
    // guide's per-CU HBM rate, with one stage already; what is left of the step is launch + merge.)
    using Row = KvRow<T, EPL>;
    typename Row::Raw k0[KPI], v0[KPI];
    auto issue = [&](int base, typename Row::Raw(&kr)[KPI], typename Row::Raw(&vr)[KPI]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < KPI; ++i) {
            const int key = min(base + i, c1 - 1); // (a key past the chunk re-reads its last row; its weight is forced to 0 below)
            const bool fresh = key == n - 1;
            kr[i] = Row::ldraw(fresh ? kn + (long)bh * D + e0 : kcache + (long)key * D + e0);
            vr[i] = Row::ldraw(fresh ? vn + (long)bh * D + e0 : vcache + (long)key * D + e0);
        }
    };
    auto consume = [&](int base, const typename Row::Raw(&kr)[KPI], const typename Row::Raw(&vr)[KPI]) __attribute__((always_inline)) {
        float sc[KPI];
#pragma unroll
        for (int i = 0; i < KPI; ++i) {
            float kf[EPL];
            Row::cvt(kr[i], kf);
            float dot = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e)
                dot = fmaf(qv[e], kf[e], dot);
            dot += __shfl_xor(dot, 1);
            dot += __shfl_xor(dot, 2);
            dot += __shfl_xor(dot, 4);
            dot += __shfl_xor(dot, 8);
            sc[i] = base + i < c1 ? dot : -INFINITY;
        }
        float m_new = m;
#pragma unroll
        for (int i = 0; i < KPI; ++i)
            m_new = fmaxf(m_new, sc[i]);
        // (the first key of the iteration is always live, so m_new is finite here)
        const float alpha = expf(m - m_new);
        l *= alpha;
#pragma unroll
        for (int e = 0; e < EPL; ++e)
            o[e] *= alpha;
#pragma unroll
        for (int i = 0; i < KPI; ++i) {
            const float pv = expf(sc[i] - m_new); // exp(-inf) = 0 for the keys past the chunk
            l += pv;
            float vf[EPL];
            Row::cvt(vr[i], vf);
#pragma unroll
            for (int e = 0; e < EPL; ++e)
                o[e] = fmaf(pv, vf[e], o[e]);
        }
        m = m_new;
    };
    for (int base = c0 + grp * KPI; base < c1; base += step) {
        issue(base, k0, v0);
        consume(base, k0, v0);
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
        for (int gg = 0; gg < NG; ++gg)
            mm = fmaxf(mm, s_m[gg]);
        float ll = 0.f, oo = 0.f;
#pragma unroll
        for (int gg = 0; gg < NG; ++gg) {
            const float wgt = s_m[gg] == -INFINITY ? 0.f : expf(s_m[gg] - mm);
            ll += s_l[gg] * wgt;
            oo += s_o[gg][t] * wgt;
        }
        if (G == 1) {
            KvLd<T>::st(out + (long)bh * D + t, oo / ll);
        } else {
            float *pp = part + ((long)bh * G + g) * (D + 2);
            pp[t] = oo;
            if (t == 0) {
                pp[D] = mm; // -inf for an empty chunk (n shorter than g chunks): the merge gives it weight 0
                pp[D + 1] = ll;
            }
        }
    }
}

// G <= 64: wave 0 reads the G (m, l) pairs in ONE round trip (lane g), reduces them with shuffles and leaves the G weights in LDS;
// every thread then sums its column over the chunks with the loads of eight chunks in flight. (The first version walked the chunks
// in two dependent loops: two memory round trips per chunk, 5.4 us for G = 8 — a quarter of the whole decode step.)
template <typename T, typename P, int D>
__global__ __launch_bounds__(D) void attention_kvcache_merge_kernel(const float *__restrict__ part, const P *__restrict__ pos, T *__restrict__ out,
                                                                      int max_seq, int G) {
    __shared__ float s_w[64], s_l;
    const int bh = blockIdx.x, t = threadIdx.x;
    const int n = (int)pos[0] + 1;
    if (n < 1 || n > max_seq)
        return; // (as the split kernel: a position outside the cache writes nothing)
    const float *pp = part + (long)bh * G * (D + 2);
    if (t < 64) {
        const float mg = t < G ? pp[(long)t * (D + 2) + D] : -INFINITY;
        const float lg = t < G ? pp[(long)t * (D + 2) + D + 1] : 0.f;
        float mm = mg;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1)
            mm = fmaxf(mm, __shfl_xor(mm, o));
        const float wgt = mg == -INFINITY ? 0.f : expf(mg - mm);
        float ll = lg * wgt;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1)
            ll += __shfl_xor(ll, o);
        s_w[t] = wgt;
        if (t == 0)
            s_l = ll;
    }
    __syncthreads();
    float oo = 0.f;
    int g = 0;
    for (; g + 8 <= G; g += 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            v[i] = pp[(long)(g + i) * (D + 2) + t];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            oo = fmaf(v[i], s_w[g + i], oo);
    }
    for (; g < G; ++g)
        oo = fmaf(pp[(long)g * (D + 2) + t], s_w[g], oo);
    KvLd<T>::st(out + (long)bh * D + t, oo / s_l);
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
    // Split of the cache over workgroups: G chunks so that batch_heads x G is ~1 workgroup per CU (measured, B x H = 32, 4096 keys, f16:
    // G = 4 / 8 / 16 / 32 -> 27 / 20.6 / 25 / 33 us; B x H = 256: G = 1 best), each chunk >= 256 keys of the
    // cache's capacity (n itself lives on the device). Vector loads need 16-byte aligned rows; anything else keeps the element-wise kernel.
    const bool vec_ok = ((((uintptr_t)k_cache) | ((uintptr_t)v_cache) | ((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v)) & 15) == 0;
    int G = (int)std::min<int64_t>(std::max<int64_t>(1, (int64_t)rt->num_cu / batch_heads), std::max<int64_t>(1, max_seq / 256));
    if (G > 64)
        G = 64;
    if (const char *force = getenv("IROCM_KVCACHE_SPLIT")) { // test hook (read per call): force G; 0 = the element-wise one-workgroup kernel
        const int vforce = atoi(force);
        G = vforce < 0 ? 0 : (vforce > 64 ? 64 : vforce); // (the merge kernel holds 64 chunk weights)
    }
    // (512-thread workgroups — 32 key groups, 128 keys per iteration — were measured: 18.8 us at B x H = 32, 4096 keys with either size.
    // So was the merge inside the split kernel — the workgroup that bumps a per-(batch, head) counter to G merges, with agent-scope
    // release / acquire fences around the counter because the partial results cross XCDs: 26.6 instead of 18.7 us at that shape, 34.5
    // instead of 19.1 at B x H = 8, 8192 keys, D = 256 — every writer's release is an L2 write-back. Two launches it stays.)
    float *part = nullptr;
    if (vec_ok && G > 1) {
        void *ws = nullptr;
        const int st = infini_rocm_workspace(rt, (size_t)batch_heads * G * (head_dim + 2) * sizeof(float), &ws);
        if (st != INFINI_ROCM_OK)
            return st;
        part = (float *)ws;
    }
#define GO(T, P, E)                                                                                        \
    if (vec_ok && G >= 1) {                                                                                \
        hipLaunchKernelGGL((attention_kvcache_split_kernel<T, P, E, 4, 16>), dim3((unsigned)batch_heads, (unsigned)G), dim3(256), 0, \
                               rt->stream, (T *)k_cache, (T *)v_cache, (const T *)q, (const T *)k, (const T *)v, \
                               (const P *)position_id, (T *)out, part, (int)max_seq, G);                   \
        if (G > 1)                                                                                         \
            hipLaunchKernelGGL((attention_kvcache_merge_kernel<T, P, E * 16>), dim3((unsigned)batch_heads), dim3(E * 16), 0, \
                               rt->stream, part, (const P *)position_id, (T *)out, (int)max_seq, G);        \
    } else                                                                                                 \
        hipLaunchKernelGGL((attention_kvcache_kernel<T, P, E>), dim3((unsigned)batch_heads), dim3(256), 0,  \
                           rt->stream, (T *)k_cache, (T *)v_cache, (const T *)q, (const T *)k, (const T *)v, \
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
