// Resize (ONNX Resize-13: nearest / linear / cubic, five coordinate transforms) for gfx950.
// Replaces _resize_kernel_nearest / _resize_kernel_coeff<2,16> / <4,256> (reference: src/kernels/cuda/resize.cu:6-196,
// glue resize.cc:5-50; operator attributes src/operators/resize.cc:30-230):
//   source coordinate of output index i along a resized dim, scale s = out / in (as computed by the operator):
//     half_pixel (i + .5) / s - .5 | pytorch_half_pixel (same, 0 when out == 1) | align_corners i (in-1)/(out-1) |
//     asymmetric i / s | tf_crop_and_resize roi-driven
//   nearest: round per nearest_mode, clamp to the input; linear / cubic (A = -0.75): separable over the resized dims,
//   neighbours clamped to the edge (no extrapolation value), weights multiplied across dims.
// Deviations: the reference's round_prefer_floor / round_prefer_ceil are floor(x + 0.4) / floor(x + 0.5)
// (resize.cu:7-13) — the first differs from ONNX for fractions in (0.5, 0.6); here ceil(x - 0.5) / floor(x + 0.5).
// The reference is fp32-only and rank <= 4; here f32 / f16 / bf16 (fp32 math), rank <= 8. Dims whose extent does not
// change take no part in the interpolation (the reference multiplies by weights 1 and 0 there — same value).
// One thread per output element; HBM-bound: reads <= N^k inputs (cache-resident neighbours), writes one output.
#include "common.h"

namespace irocm {

template <typename T> struct RsLd;
template <> struct RsLd<float> {
    __device__ static inline float ld(const float *p) { return *p; }
    __device__ static inline void st(float *p, float v) { *p = v; }
};
template <> struct RsLd<__half> {
    __device__ static inline float ld(const __half *p) { return __half2float(*p); }
    __device__ static inline void st(__half *p, float v) { *p = __float2half_rn(v); }
};
template <> struct RsLd<__hip_bfloat16> {
    __device__ static inline float ld(const __hip_bfloat16 *p) { return __bfloat162float(*p); }
    __device__ static inline void st(__hip_bfloat16 *p, float v) { *p = __float2bfloat16(v); }
};

struct ResizeArgs {
    int ndim, nrd;                       // rank; number of resized dims
    long total;
    int in_dims[INFINI_ROCM_MAX_DIMS], out_dims[INFINI_ROCM_MAX_DIMS];
    long in_stride[INFINI_ROCM_MAX_DIMS];
    float scale[INFINI_ROCM_MAX_DIMS], roi_s[INFINI_ROCM_MAX_DIMS], roi_e[INFINI_ROCM_MAX_DIMS];
    int coord_mode, nearest_mode;
};

__device__ inline float src_coord(int idx, const ResizeArgs &p, int d) {
    const float s = p.scale[d];
    const float len = s * (float)p.in_dims[d]; // "resizedLen" of the reference
    switch (p.coord_mode) {
    case 0: return ((float)idx + 0.5f) / s - 0.5f;
    case 1: return len > 1.f ? ((float)idx + 0.5f) / s - 0.5f : 0.f;
    case 2: return len == 1.f ? 0.f : (float)idx * (float)(p.in_dims[d] - 1) / (len - 1.f);
    case 3: return (float)idx / s;
    default: {
        const int li = (int)len;
        return li > 1 ? p.roi_s[d] * (float)(p.in_dims[d] - 1) +
                            (float)idx * (p.roi_e[d] - p.roi_s[d]) * (float)(p.in_dims[d] - 1) / (float)(li - 1)
                      : 0.5f * (p.roi_s[d] + p.roi_e[d]) * (float)(p.in_dims[d] - 1);
    }
    }
}

__device__ inline int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

template <typename T>
__global__ __launch_bounds__(256) void resize_nearest_kernel(const T *__restrict__ x, T *__restrict__ y, ResizeArgs p) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long)gridDim.x * 256) {
        long rem = i, off = 0;
        for (int d = p.ndim - 1; d >= 0; --d) {
            const long q = rem / p.out_dims[d];
            const int oi = (int)(rem - q * p.out_dims[d]);
            rem = q;
            int si = oi;
            if (p.in_dims[d] != p.out_dims[d]) {
                const float c = src_coord(oi, p, d);
                switch (p.nearest_mode) {
                case 0: si = (int)ceilf(c - 0.5f); break;  // round_prefer_floor
                case 1: si = (int)floorf(c + 0.5f); break; // round_prefer_ceil
                case 2: si = (int)floorf(c); break;
                default: si = (int)ceilf(c); break;
                }
                si = clampi(si, p.in_dims[d] - 1);
            }
            off += (long)si * p.in_stride[d];
        }
        y[i] = x[off];
    }
}

// N = 2 (linear) or 4 (cubic) neighbours per resized dim
template <typename T, int N>
__global__ __launch_bounds__(256) void resize_interp_kernel(const T *__restrict__ x, T *__restrict__ y, ResizeArgs p) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long)gridDim.x * 256) {
        long rem = i, base = 0;
        long noff[4][N]; // per resized dim (at most 4 interpolated dims): neighbour offsets and weights
        float nw[4][N];
        int k = 0;
        for (int d = p.ndim - 1; d >= 0; --d) {
            const long q = rem / p.out_dims[d];
            const int oi = (int)(rem - q * p.out_dims[d]);
            rem = q;
            if (p.in_dims[d] == p.out_dims[d]) {
                base += (long)oi * p.in_stride[d];
                continue;
            }
            const float c = src_coord(oi, p, d);
            const int fl = (int)floorf(c);
            const float r = c - (float)fl;
            float w[4];
            if (N == 2) {
                w[0] = 1.f - r;
                w[1] = r;
            } else {
                const float A = -0.75f;
                w[0] = ((A * (r + 1) - 5 * A) * (r + 1) + 8 * A) * (r + 1) - 4 * A;
                w[1] = ((A + 2) * r - (A + 3)) * r * r + 1;
                w[2] = ((A + 2) * (1 - r) - (A + 3)) * (1 - r) * (1 - r) + 1;
                w[3] = ((A * ((1 - r) + 1) - 5 * A) * ((1 - r) + 1) + 8 * A) * ((1 - r) + 1) - 4 * A;
            }
#pragma unroll
            for (int n = 0; n < N; ++n) {
                noff[k][n] = (long)clampi(fl - N / 2 + 1 + n, p.in_dims[d] - 1) * p.in_stride[d];
                nw[k][n] = w[n];
            }
            ++k;
        }
        int combos = 1;
        for (int j = 0; j < k; ++j)
            combos *= N;
        float acc = 0.f;
        for (int cidx = 0; cidx < combos; ++cidx) {
            int c = cidx;
            long off = base;
            float wt = 1.f;
            for (int j = 0; j < k; ++j) {
                const int n = c % N;
                c /= N;
                off += noff[j][n];
                wt *= nw[j][n];
            }
            acc += RsLd<T>::ld(x + off) * wt;
        }
        RsLd<T>::st(y + i, acc);
    }
}

} // namespace irocm

using namespace irocm;

extern "C" int infini_rocm_resize(infiniRocmRuntime_t rt, int dtype, const void *x, void *y, int ndim,
                                  const int64_t *in_shape, const int64_t *out_shape, const float *scales,
                                  const float *roi, int mode, int coord_mode, int nearest_mode) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(ndim >= 1 && ndim <= INFINI_ROCM_MAX_DIMS && in_shape && out_shape && scales, "resize: bad rank / NULL shape");
    IROCM_CHECK_ARG(mode >= 0 && mode <= 2 && coord_mode >= 0 && coord_mode <= 4 && nearest_mode >= 0 && nearest_mode <= 3,
                    "resize: bad mode (%d, %d, %d)", mode, coord_mode, nearest_mode);
    IROCM_CHECK_ARG(coord_mode != 4 || roi, "resize: tf_crop_and_resize needs roi");
    ResizeArgs p;
    p.ndim = ndim;
    p.nrd = 0;
    p.total = 1;
    long st = 1;
    for (int d = ndim - 1; d >= 0; --d) {
        IROCM_CHECK_ARG(in_shape[d] > 0 && out_shape[d] >= 0 && in_shape[d] < (1ll << 31) && out_shape[d] < (1ll << 31),
                        "resize: bad extent");
        p.in_dims[d] = (int)in_shape[d];
        p.out_dims[d] = (int)out_shape[d];
        p.in_stride[d] = st;
        st *= in_shape[d];
        p.total *= out_shape[d];
        p.scale[d] = scales[d];
        p.roi_s[d] = roi ? roi[d] : 0.f;
        p.roi_e[d] = roi ? roi[d + ndim] : 1.f;
        if (in_shape[d] != out_shape[d])
            ++p.nrd;
    }
    IROCM_CHECK_ARG(mode == 0 || p.nrd <= 4, "resize: at most 4 interpolated dims");
    p.coord_mode = coord_mode;
    p.nearest_mode = nearest_mode;
    if (p.total == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(x && y, "resize: NULL tensor");
    long g = ceil_div(p.total, 256);
    if (g > (long)rt->num_cu * 16) g = (long)rt->num_cu * 16;
#define GO(T)                                                                                                  \
    if (mode == 0)                                                                                             \
        hipLaunchKernelGGL(resize_nearest_kernel<T>, dim3((unsigned)g), dim3(256), 0, rt->stream, (const T *)x, (T *)y, p); \
    else if (mode == 1)                                                                                        \
        hipLaunchKernelGGL((resize_interp_kernel<T, 2>), dim3((unsigned)g), dim3(256), 0, rt->stream, (const T *)x, (T *)y, p); \
    else                                                                                                       \
        hipLaunchKernelGGL((resize_interp_kernel<T, 4>), dim3((unsigned)g), dim3(256), 0, rt->stream, (const T *)x, (T *)y, p)
    switch (dtype) {
    case INFINI_DT_F32: GO(float); break;
    case INFINI_DT_F16: GO(__half); break;
    case INFINI_DT_BF16: GO(__hip_bfloat16); break;
    default: IROCM_FAIL(INFINI_ROCM_UNSUPPORTED, "resize: unsupported dtype %s", dtype_name(dtype));
    }
#undef GO
    IROCM_LAUNCH_CHECK("resize");
    return INFINI_ROCM_OK;
}
