// Conv2d for gfx950: implicit-GEMM on MFMA, NCHW x FCRS -> NFHW, no layout transform in HBM.
//
// Replaces convCudnn (reference: src/kernels/cuda/conv.cc:57-168, cudnnConvolutionForward, default
// algo IMPLICIT_GEMM); op definition src/operators/conv.cc:47-114; index math as the native CPU
// kernel src/kernels/cpu/conv.cc:25-50 (cross-correlation, symmetric zero padding, stride, dilation,
// groups = C / channel_per_group). The reference op has no bias and its CUDA kernel ignores `act`;
// `bias` ([F], optional) and `act` are here so a runtime can fuse the Conv -> Add(bias) -> Relu chain
// the ONNX front-end emits (pyinfinitensor onnx.py:159-190) without changing results.
//
// GEMM view, per image n and group g:   Y[n, g*Fg + m, p] = sum_k W[g*Fg + m, k] * B[k, p]
//   m in [0, Fg), p = oh*OW + ow in [0, OH*OW), k = (c*R + r)*S + s in [0, Cg*R*S)
//   B[k, p] = X[n, g*Cg + c, oh*sh - ph + r*dh, ow*sw - pw + s*dw]  (0 outside the image)
// W rows are K-major and contiguous (FCRS), B is pixel-major: exactly the "A K-major, B N-major"
// case of gemm.hip, with B gathered on the fly.
//
// conv_igemm16 (f16 / bf16): 128(filters) x 128(pixels) x 32(k) tile, 4 waves (2x2, 64x64 each),
//   v_mfma_f32_16x16x32, fp32 accumulate. A tile -> LDS [128][40] (16-byte ds_read_b128 fragments,
//   80-byte pitch = conflict-free); B tile -> LDS [32 k][128 p] written as 16-byte pixel runs and read
//   with ds_read_b64_tr_b16 (transpose read), XOR-swizzled like gemm.hip. Global loads of tile t+1
//   are issued before the MFMAs of tile t and written to LDS after them (register-staged pipeline).
//   1x1 / stride 1 / pad 0 convolutions never come here: they are plain batched GEMMs and are routed to
//   infini_rocm_matmul's LDS-DMA kernels by the dispatcher when the plane size allows 16-byte rows.
// conv_direct32 (f32): one output per thread, serial fp32 fma over k in the oracle's order.
#include "gemm_common.h"
#include <type_traits>

extern "C" int infini_rocm_matmul(infiniRocmRuntime_t rt, int dtype, const void *a, const void *b,
                                  const void *bias, void *c, int64_t batch, int64_t m, int64_t n,
                                  int64_t k, int trans_a, int trans_b, int64_t stride_a, int64_t stride_b,
                                  int64_t bias_stride_b, int64_t bias_stride_m, int64_t bias_stride_n,
                                  int act);

namespace irocm {

template <typename T> struct CvtT;
template <> struct CvtT<float> {
    __device__ static inline float ld(const float *p) { return *p; }
    __device__ static inline void st(float *p, float v) { *p = v; }
};
template <> struct CvtT<__half> {
    __device__ static inline float ld(const __half *p) { return __half2float(*p); }
    __device__ static inline void st(__half *p, float v) { *p = __float2half_rn(v); }
};
template <> struct CvtT<__hip_bfloat16> {
    __device__ static inline float ld(const __hip_bfloat16 *p) { return __bfloat162float(*p); }
    __device__ static inline void st(__hip_bfloat16 *p, float v) { *p = __float2bfloat16(v); }
};

int launch_conv_s1(infiniRocmRuntime_t rt, int dtype, const void *x, const void *w, const void *bias, const void *res,
                   void *y, int n, int c, int h, int wd, int f, int r, int s, int ph, int pw, int sh, int sw, int dh, int dw,
                   int oh, int ow, int act); // conv_s1.hip

int launch_conv_depthwise(infiniRocmRuntime_t rt, int dtype, const void *x, const void *w, const void *bias, void *y, int64_t n, int64_t c,
                          int64_t h, int64_t wd, int64_t f, int r, int s, int ph, int pw, int sh, int sw, int oh, int ow, int act); // conv_dw.hip

int launch_conv_igemm32(infiniRocmRuntime_t rt, const void *x, const void *w, const void *bias, const void *res, void *y, int64_t n,
                        int64_t c, int64_t h, int64_t wd, int64_t f, int r, int s, int ph, int pw, int sh, int sw, int dh, int dw, int oh,
                        int ow, int act); // gemm32.hip

struct ConvArgs {
    const void *x, *w, *bias, *res; // res: optional residual of y's shape, added before the activation
    void *y;
    int n, c, h, wd, f, r, s;
    int ph, pw, sh, sw, dh, dw;
    int groups, cg, fg; // channels / filters per group
    int oh, ow;
    int kdim;           // cg * r * s
    int npix;           // oh * ow
    int tiles_m, tiles_p;
    int act;
    unsigned magic_s;   // (65536 + s - 1) / s : rs / s == (rs * magic_s) >> 16 for rs < 4096
};

template <typename Tr> __global__ __launch_bounds__(256) void conv_igemm16(ConvArgs p) {
    constexpr int BM = 128, BP = 128, BK = 32, APITCH = BK + 8;
    __shared__ __attribute__((aligned(16))) unsigned short As[BM][APITCH]; // 10 KiB
    __shared__ __attribute__((aligned(16))) unsigned short Bs[BK * BP];    // 8 KiB, [32][128] swizzled
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1;

    // block -> (image, group, filter tile, pixel tile); pixel tiles fastest so neighbours share weights
    unsigned bid = blockIdx.x;
    const int tp = bid % p.tiles_p; bid /= p.tiles_p;
    const int tm = bid % p.tiles_m; bid /= p.tiles_m;
    const int g = bid % p.groups;
    const int img = bid / p.groups;
    const int m0 = tm * BM, p0 = tp * BP;

    const unsigned short *W = (const unsigned short *)p.w + (long)(g * p.fg) * p.kdim;
    const unsigned short *X = (const unsigned short *)p.x + ((long)img * p.c + (long)g * p.cg) * p.h * p.wd;
    const int RS = p.r * p.s;
    // bias of the lane's four filter rows, loaded now: a 2-byte load issued in the epilogue puts an L2 round trip in front
    // of the tile's stores (measured on conv_s1.hip: the epilogue, not the K loop, bounded the pointwise layers)
    float bias_v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int fm = m0 + wm * 64 + i * 16 + (lane & 15);
        bias_v[i] = (p.bias && fm < p.fg) ? Tr::to_f32(((const unsigned short *)p.bias)[g * p.fg + fm]) : 0.f;
    }

    // ---- per-thread staging assignment --------------------------------------------------------
    // A: 128 rows x 32 k = 512 chunks of 8 k; thread t takes chunks t and t + 256: row = ch >> 2, kc = ch & 3
    // B: 32 k x 128 p = 512 chunks of 8 pixels; thread t takes chunks t and t + 256: kk = ch >> 4, pc = ch & 15
    const bool a_vec = (p.kdim % 8 == 0) && ((((uintptr_t)p.w) & 15) == 0);
    int b_oh[2], b_ow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ch = t + i * 256;
        const int pix = p0 + (ch & 15) * 8;
        b_oh[i] = pix / p.ow;
        b_ow[i] = pix - b_oh[i] * p.ow;
    }
    s16x8_t a_reg[2], b_reg[2];

    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = t + i * 256;
            const int row = ch >> 2, kc = (ch & 3) * 8;
            const int gm = m0 + row, gk = k0 + kc;
            s16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (gm < p.fg) {
                const unsigned short *src = W + (long)gm * p.kdim + gk;
                if (a_vec && gk + 8 <= p.kdim) {
                    v = *(const s16x8_t *)src;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (gk + j < p.kdim)
                            v[j] = (short)src[j];
                }
            }
            a_reg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = t + i * 256;
            const int k = k0 + (ch >> 4);
            s16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (k < p.kdim) {
                const int cc = k / RS;
                const int rs = k - cc * RS;
                const int rr = (int)(((unsigned)rs * p.magic_s) >> 16);
                const int ss = rs - rr * p.s;
                const unsigned short *plane = X + (long)cc * p.h * p.wd;
                int oh = b_oh[i], ow = b_ow[i];
                const int ih0 = oh * p.sh - p.ph + rr * p.dh;
                const int iw0 = ow * p.sw - p.pw + ss * p.dw;
                const int pix = p0 + (ch & 15) * 8;
                // fast path: 8 pixels in one output row, unit stride, all inside the image, 16-B aligned
                if (p.sw == 1 && ow + 8 <= p.ow && pix + 8 <= p.npix && ih0 >= 0 && ih0 < p.h && iw0 >= 0 &&
                    iw0 + 8 <= p.wd) {
                    const unsigned short *src = plane + (long)ih0 * p.wd + iw0;
                    if ((((uintptr_t)src) & 15) == 0) {
                        v = *(const s16x8_t *)src;
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            v[j] = (short)src[j];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (pix + j < p.npix) {
                            const int ih = oh * p.sh - p.ph + rr * p.dh;
                            const int iw = ow * p.sw - p.pw + ss * p.dw;
                            if (ih >= 0 && ih < p.h && iw >= 0 && iw < p.wd)
                                v[j] = (short)plane[(long)ih * p.wd + iw];
                        }
                        if (++ow == p.ow) {
                            ow = 0;
                            ++oh;
                        }
                    }
                }
            }
            b_reg[i] = v;
        }
    };

    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = t + i * 256;
            *(s16x8_t *)&As[ch >> 2][(ch & 3) * 8] = a_reg[i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = t + i * 256;
            const int kk = ch >> 4, pc = ch & 15;
            const int c16 = pc ^ (f128::mn_f(kk) << 1);
            *(s16x8_t *)((char *)Bs + kk * 256 + c16 * 16) = b_reg[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.kdim + BK - 1) / BK;
    load_tile(0);
    for (int kt = 0; kt < nk; ++kt) {
        store_tile();
        __syncthreads();
        if (kt + 1 < nk)
            load_tile((kt + 1) * BK); // in flight during the MFMAs below
        s16x8_t af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            af[i] = *(const s16x8_t *)&As[wm * 64 + i * 16 + (lane & 15)][(lane >> 4) * 8];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            bf[j] = f128::frag_mnmajor((const char *)Bs, wn * 64 + j * 16, 0, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = Tr::mfma(bf[j], af[i], acc[i][j]); // swapped: lane holds 4 consecutive pixels
        __syncthreads();
    }

    unsigned short *Y = (unsigned short *)p.y + ((long)img * p.f + (long)g * p.fg) * p.npix;
    const unsigned short *bias = (const unsigned short *)p.bias;
    const bool vec_ok = (p.npix % 4 == 0) && ((((uintptr_t)p.y) & 7) == 0);
    // wide path: 16-byte stores by swapping half tiles between lane groups g4 / g4^1 (see gemm256.hip / conv_s1.hip)
    const bool wide = (p.npix % 16 == 0) && ((((uintptr_t)p.y) & 15) == 0) && (p0 + wn * 64 + 64 <= p.npix) &&
                      (m0 + wm * 64 + 64 <= p.fg) && (!p.res || ((((uintptr_t)p.res) & 7) == 0));
    if (wide) {
        const int l15 = lane & 15, g4 = lane >> 4;
        const bool odd = g4 & 1;
        // one copy per activation (a runtime switch per element leaves 64 scalar branches in the loop)
        auto store_all = [&](auto actc) __attribute__((always_inline)) {
        constexpr int ACT = decltype(actc)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int fm = m0 + wm * 64 + i * 16 + l15;
            const float bv = bias_v[i]; // loaded before the K loop
            const long rowoff = (long)fm * p.npix;
            const unsigned short *R = p.res ? (const unsigned short *)p.res + ((long)img * p.f + (long)g * p.fg) * p.npix : nullptr;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                unsigned pk[2][2];
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    const int pix = p0 + wn * 64 + (jp * 2 + t2) * 16 + g4 * 4;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        v[r] = acc[i][jp * 2 + t2][r] + bv;
                    if (R) {
                        const u32x2_t rk = *(const u32x2_t *)(R + rowoff + pix);
                        v[0] += Tr::to_f32((unsigned short)(rk[0] & 0xffff)); v[1] += Tr::to_f32((unsigned short)(rk[0] >> 16));
                        v[2] += Tr::to_f32((unsigned short)(rk[1] & 0xffff)); v[3] += Tr::to_f32((unsigned short)(rk[1] >> 16));
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (ACT == 1)
                            v[r] = v[r] > 0.f ? v[r] : 0.f;
                        else if constexpr (ACT < 0)
                            v[r] = apply_act(v[r], p.act);
                    }
                    pk[t2][0] = (unsigned)Tr::from_f32(v[0]) | ((unsigned)Tr::from_f32(v[1]) << 16);
                    pk[t2][1] = (unsigned)Tr::from_f32(v[2]) | ((unsigned)Tr::from_f32(v[3]) << 16);
                }
                const unsigned s0 = odd ? pk[0][0] : pk[1][0], s1 = odd ? pk[0][1] : pk[1][1];
                const unsigned r0 = (unsigned)__shfl_xor((int)s0, 16), r1 = (unsigned)__shfl_xor((int)s1, 16);
                u32x4_t o;
                if (odd) { o[0] = r0; o[1] = r1; o[2] = pk[1][0]; o[3] = pk[1][1]; }
                else { o[0] = pk[0][0]; o[1] = pk[0][1]; o[2] = r0; o[3] = r1; }
                const int pix = p0 + wn * 64 + (jp * 2 + (odd ? 1 : 0)) * 16 + (g4 & ~1) * 4;
                *(u32x4_t *)(Y + rowoff + pix) = o;
            }
        }
        };
        if (p.act == 0)
            store_all(std::integral_constant<int, 0>{});
        else if (p.act == 1)
            store_all(std::integral_constant<int, 1>{});
        else
            store_all(std::integral_constant<int, -1>{});
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int fm = m0 + wm * 64 + i * 16 + (lane & 15);
        if (fm >= p.fg)
            continue;
        const float bv = bias ? Tr::to_f32(bias[g * p.fg + fm]) : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pix = p0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            if (pix >= p.npix)
                continue;
            float v[4];
            const unsigned short *rp = p.res ? (const unsigned short *)p.res + ((long)img * p.f + (long)g * p.fg + fm) * p.npix + pix : nullptr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = acc[i][j][r] + bv;
                if (rp && pix + r < p.npix)
                    t += Tr::to_f32(rp[r]);
                v[r] = apply_act(t, p.act);
            }
            unsigned short *dst = Y + (long)fm * p.npix + pix;
            if (vec_ok && pix + 3 < p.npix) {
                u32x2_t pk;
                pk[0] = (unsigned)Tr::from_f32(v[0]) | ((unsigned)Tr::from_f32(v[1]) << 16);
                pk[1] = (unsigned)Tr::from_f32(v[2]) | ((unsigned)Tr::from_f32(v[3]) << 16);
                *(u32x2_t *)dst = pk;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (pix + r < p.npix)
                        dst[r] = Tr::from_f32(v[r]);
            }
        }
    }
}

// fp32: exact-order fma chain per output (c, then r, then s — the loop nest of src/kernels/cpu/conv.cc)
__global__ __launch_bounds__(256) void conv_direct32(ConvArgs p) {
    const long total = (long)p.n * p.f * p.npix;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int pix = (int)(i % p.npix);
        const int f = (int)((i / p.npix) % p.f);
        const int img = (int)(i / ((long)p.npix * p.f));
        const int g = f / p.fg;
        const int oh = pix / p.ow, ow = pix - oh * p.ow;
        const float *X = (const float *)p.x + ((long)img * p.c + (long)g * p.cg) * p.h * p.wd;
        const float *W = (const float *)p.w + (long)f * p.kdim;
        float acc = 0.f;
        for (int c = 0; c < p.cg; ++c)
            for (int r = 0; r < p.r; ++r) {
                const int ih = oh * p.sh - p.ph + r * p.dh;
                if (ih < 0 || ih >= p.h)
                    continue;
                for (int s = 0; s < p.s; ++s) {
                    const int iw = ow * p.sw - p.pw + s * p.dw;
                    if (iw < 0 || iw >= p.wd)
                        continue;
                    acc = fmaf(X[((long)c * p.h + ih) * p.wd + iw], W[(c * p.r + r) * p.s + s], acc);
                }
            }
        if (p.bias)
            acc += ((const float *)p.bias)[f];
        if (p.res)
            acc += ((const float *)p.res)[i];
        ((float *)p.y)[i] = apply_act(acc, p.act);
    }
}

// ConvTranspose2d (gather form): one output element per thread, fp32 accumulation in the (f, r, s) order.
// Replaces convBackwardDataCudnn (reference: src/kernels/cuda/conv_transposed.cc:46-230); shape rule
// src/operators/conv.cc:252-268: x [N, F, H, W], w [F, C/g, R, S] -> y [N, C, OH, OW],
//   OH = (H - 1) sh - 2 ph + dh (R - 1) + oph + 1;  y[n, c, oy, ox] = sum_{f, r, s} x[n, f, iy, ix] w[f, c, r, s]
//   with iy * sh = oy + ph - r * dh (terms whose iy is fractional or outside the input are absent).
// Functional coverage of SURVEY 8f-4 (not a tuned kernel: no MFMA path yet).
struct ConvTArgs {
    int n, f, h, w, c, r, s, ph, pw, sh, sw, dh, dw, groups, fg, cg, oh, ow, act;
};
template <typename T>
__global__ __launch_bounds__(256) void conv_transpose_direct(const T *__restrict__ x, const T *__restrict__ w,
                                                             const T *__restrict__ bias, T *__restrict__ y, ConvTArgs p) {
    const long total = (long)p.n * p.c * p.oh * p.ow;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ox = (int)(i % p.ow);
        long q = i / p.ow;
        const int oy = (int)(q % p.oh);
        q /= p.oh;
        const int c = (int)(q % p.c), n = (int)(q / p.c);
        const int g = c / p.cg, cl = c - g * p.cg;
        float acc = 0.f;
        for (int fl = 0; fl < p.fg; ++fl) {
            const int f = g * p.fg + fl;
            const T *xp = x + ((long)n * p.f + f) * p.h * p.w;
            const T *wp = w + ((long)f * p.cg + cl) * p.r * p.s;
            for (int r = 0; r < p.r; ++r) {
                const int ty = oy + p.ph - r * p.dh;
                if (ty < 0 || ty % p.sh)
                    continue;
                const int iy = ty / p.sh;
                if (iy >= p.h)
                    continue;
                for (int s = 0; s < p.s; ++s) {
                    const int tx = ox + p.pw - s * p.dw;
                    if (tx < 0 || tx % p.sw)
                        continue;
                    const int ix = tx / p.sw;
                    if (ix >= p.w)
                        continue;
                    acc = fmaf(CvtT<T>::ld(xp + (long)iy * p.w + ix), CvtT<T>::ld(wp + r * p.s + s), acc);
                }
            }
        }
        if (bias)
            acc += CvtT<T>::ld(bias + c);
        CvtT<T>::st(y + i, apply_act(acc, p.act));
    }
}

} // namespace irocm

using namespace irocm;

namespace irocm {
// gemm256p_conv.hip: a unit-stride pointwise convolution as one GEMM over pixel slots; -1 when the operands do not qualify
int launch_conv_pw_gemm(infiniRocmRuntime_t rt, int dtype, const void *x, const void *w, const void *bias, const void *res, void *y,
                        int64_t n, int64_t c, int64_t hw, int64_t f, int act);
} // namespace irocm

extern "C" {

int infini_rocm_conv_transpose2d(infiniRocmRuntime_t rt, int dtype, const void *x, const void *w, const void *bias,
                                 void *y, int64_t n, int64_t f, int64_t h, int64_t wd, int64_t c_per_group, int64_t r,
                                 int64_t s, int ph, int pw, int sh, int sw, int dh, int dw, int oph, int opw,
                                 int64_t groups, int act) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(dtype == INFINI_DT_F32 || dtype == INFINI_DT_F16 || dtype == INFINI_DT_BF16,
                    "conv_transpose2d: unsupported dtype %s", dtype_name(dtype));
    IROCM_CHECK_ARG(n >= 0 && f > 0 && h > 0 && wd > 0 && c_per_group > 0 && r > 0 && s > 0, "conv_transpose2d: bad extent");
    IROCM_CHECK_ARG(groups > 0 && f % groups == 0, "conv_transpose2d: groups %lld do not divide F=%lld", (long long)groups,
                    (long long)f);
    IROCM_CHECK_ARG(sh > 0 && sw > 0 && dh > 0 && dw > 0 && ph >= 0 && pw >= 0 && oph >= 0 && opw >= 0 && act >= 0 && act <= 3,
                    "conv_transpose2d: bad attributes");
    ConvTArgs p;
    p.n = (int)n; p.f = (int)f; p.h = (int)h; p.w = (int)wd; p.cg = (int)c_per_group; p.c = (int)(c_per_group * groups);
    p.r = (int)r; p.s = (int)s; p.ph = ph; p.pw = pw; p.sh = sh; p.sw = sw; p.dh = dh; p.dw = dw;
    p.groups = (int)groups; p.fg = (int)(f / groups);
    p.oh = (int)((h - 1) * sh - 2 * ph + dh * (r - 1) + oph + 1);
    p.ow = (int)((wd - 1) * sw - 2 * pw + dw * (s - 1) + opw + 1);
    p.act = act;
    IROCM_CHECK_ARG(p.oh > 0 && p.ow > 0, "conv_transpose2d: empty output %dx%d", p.oh, p.ow);
    const long total = (long)n * p.c * p.oh * p.ow;
    if (total == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(x && w && y, "conv_transpose2d: NULL tensor");
    long g = ceil_div(total, 256);
    if (g > (long)rt->num_cu * 32) g = (long)rt->num_cu * 32;
#define GO(T)                                                                                              \
    hipLaunchKernelGGL(conv_transpose_direct<T>, dim3((unsigned)g), dim3(256), 0, rt->stream, (const T *)x, \
                       (const T *)w, (const T *)bias, (T *)y, p)
    switch (dtype) {
    case INFINI_DT_F32: GO(float); break;
    case INFINI_DT_F16: GO(__half); break;
    default: GO(__hip_bfloat16); break;
    }
#undef GO
    IROCM_LAUNCH_CHECK("conv_transpose_direct");
    return INFINI_ROCM_OK;
}


int infini_rocm_conv2d(infiniRocmRuntime_t rt, int dtype, const void *x, const void *w, const void *bias,
                       void *y, int64_t n, int64_t c, int64_t h, int64_t wd, int64_t f, int64_t r, int64_t s,
                       int ph, int pw, int sh, int sw, int dh, int dw, int64_t groups, int act) {
    return infini_rocm_conv2d_res(rt, dtype, x, w, bias, nullptr, y, n, c, h, wd, f, r, s, ph, pw, sh, sw, dh, dw, groups, act);
}

int infini_rocm_conv2d_res(infiniRocmRuntime_t rt, int dtype, const void *x, const void *w, const void *bias,
                           const void *residual, void *y, int64_t n, int64_t c, int64_t h, int64_t wd, int64_t f, int64_t r,
                           int64_t s, int ph, int pw, int sh, int sw, int dh, int dw, int64_t groups, int act) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(dtype == INFINI_DT_F32 || dtype == INFINI_DT_F16 || dtype == INFINI_DT_BF16,
                    "conv2d: unsupported dtype %s", dtype_name(dtype));
    IROCM_CHECK_ARG(n >= 0 && c > 0 && h > 0 && wd > 0 && f > 0 && r > 0 && s > 0, "conv2d: bad extent");
    IROCM_CHECK_ARG(groups > 0 && c % groups == 0 && f % groups == 0, "conv2d: groups %lld do not divide C=%lld / F=%lld",
                    (long long)groups, (long long)c, (long long)f);
    IROCM_CHECK_ARG(sh > 0 && sw > 0 && dh > 0 && dw > 0 && ph >= 0 && pw >= 0, "conv2d: bad attributes");
    IROCM_CHECK_ARG(act >= 0 && act <= 3, "conv2d: bad act %d", act);
    ConvArgs p;
    p.x = x; p.w = w; p.bias = bias; p.res = residual; p.y = y;
    p.n = (int)n; p.c = (int)c; p.h = (int)h; p.wd = (int)wd; p.f = (int)f; p.r = (int)r; p.s = (int)s;
    p.ph = ph; p.pw = pw; p.sh = sh; p.sw = sw; p.dh = dh; p.dw = dw;
    p.groups = (int)groups; p.cg = (int)(c / groups); p.fg = (int)(f / groups);
    // reference output size: src/operators/conv.cc:98-101
    p.oh = (int)((h - (r - sh) * dh + 2 * ph) / sh);
    p.ow = (int)((wd - (s - sw) * dw + 2 * pw) / sw);
    IROCM_CHECK_ARG(p.oh > 0 && p.ow > 0, "conv2d: empty output %dx%d", p.oh, p.ow);
    p.kdim = p.cg * p.r * p.s;
    p.npix = p.oh * p.ow;
    p.act = act;
    p.magic_s = (65536u + (unsigned)s - 1) / (unsigned)s;
    IROCM_CHECK_ARG((long)p.r * p.s < 4096, "conv2d: kernel window too large");
    if (n == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(x && w && y, "conv2d: NULL tensor");

    if (dtype == INFINI_DT_F32) {
        // Round 5: fp32 convolutions on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32: exact products and sums at 157 TF/s): every
        // groups == 1 layer is the implicit GEMM of gemm32.hip (columns run across images; K rows that are not a multiple of 4 floats —
        // the 3-channel stem — are copied into padded rows). Unit-stride pointwise layers were first routed to the fp32 tile GEMM as
        // one GEMM per image (zero copy, "batched_gemm32"): measured at batch 32 the implicit GEMM with 64 x 64 tiles is faster on every
        // ResNet-50 layer but one (C64 -> 64 @ 56^2: 25.5 vs 69 us; C1024 -> 256 @ 14^2: 52.9 vs 101; C512 -> 256 @ 28^2: 82.6 vs 79.8) —
        // per-image GEMMs leave 23 % of a 14^2 / 7^2 plane's tiles empty and launch few workgroups; IROCM_CONV32_PW_BATCHED keeps that
        // route for A/B. conv variant 1 keeps the one-output-per-thread kernel (A/B, tests); so do grouped layers and unaligned operands.
        if (groups == 1 && rt->conv_variant != 1) {
            if (r == 1 && s == 1 && ph == 0 && pw == 0 && sh == 1 && sw == 1 && dh == 1 && dw == 1 && !residual && p.npix % 4 == 0 && c % 4 == 0 &&
                ((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)y)) & 15) == 0 && (long)n * f * p.npix < (1l << 31) &&
                getenv("IROCM_CONV32_PW_BATCHED")) {
                rt->last_conv_route = "batched_gemm32";
                return infini_rocm_matmul(rt, dtype, w, x, bias, y, n, f, p.npix, c, 0, 0, 0, (int64_t)c * p.npix, 0, bias ? 1 : 0, 0, act);
            }
            rt->last_conv_route = "igemm32"; // ("igemm32_splitk" when the launcher splits K)
            const int st = launch_conv_igemm32(rt, x, w, bias, residual, y, n, c, h, wd, f, (int)r, (int)s, ph, pw, sh, sw, dh, dw, p.oh, p.ow, act);
            if (st >= 0)
                return st;
        }
        const long total = (long)n * f * p.npix;
        long g = ceil_div(total, 256);
        if (g > (long)rt->num_cu * 32) g = (long)rt->num_cu * 32;
        hipLaunchKernelGGL(conv_direct32, dim3((unsigned)g), dim3(256), 0, rt->stream, p);
        IROCM_LAUNCH_CHECK("conv_direct32");
        rt->last_conv_route = "direct32";
        return INFINI_ROCM_OK;
    }
    const int variant = rt->conv_variant;
    // Round 5: depthwise layers (groups == C: one input channel per filter) have their own HBM-bound kernel (conv_dw.hip); variant 1
    // keeps the generic implicit GEMM (A/B, tests)
    if (groups == c && groups > 1 && dh == 1 && dw == 1 && !residual && variant != 1) {
        const int st = launch_conv_depthwise(rt, dtype, x, w, bias, y, n, c, h, wd, f, (int)r, (int)s, ph, pw, sh, sw, p.oh, p.ow, act);
        if (st >= 0)
            return st;
    }
    const bool same_s1 = sh == 1 && sw == 1 && dh == 1 && dw == 1 && groups == 1 && p.oh == p.h && p.ow == p.wd;
    // Round 3: a unit-stride pointwise layer with >= 256 filters is ONE GEMM  Y[f][slot] = W[f][c] X[c][slot]  over pixel slots
    // (image, pixel) on the persistent 256-row kernels in conv mode (gemm256p_kernel.h, CONV): LDS-DMA staging of both operands,
    // tiles that span images (14 x 14 and 7 x 7 planes do not waste tiles), per-filter bias, residual and activation in the epilogue,
    // NCHW stores. As plain GEMMs these layers run 1.2-1.8 x faster on that machinery than on the register-staged tap-shifted
    // kernel (tools/probes/conv_as_gemm.py, profiles/r03_conv_as_gemm.txt); with fewer filters the 256-row tile is mostly empty
    // and the old kernels win. Variant 5 forces it for every eligible shape.
    if ((variant < 0 || variant == 5) && r == 1 && s == 1 && ph == 0 && pw == 0 && same_s1 && c % 64 == 0 && (act == 0 || act == 1) &&
        // (>= 128 filters: C256->F128 @56x56 78 vs 100 us, C512->F128 @28x28 33 vs 40; with 64 the 256-row tile is 3/4 empty:
        // 69 vs 60 us. The grid may be thin — C1024->F256 @14x14 is 100 tiles of 256^2 and still 22.6 vs 31.1 us, C2048->F512 @7x7
        // 50 tiles and 38.8 vs 45.1 — but below ~3/16 of the CUs the 128 x 128 tiles of the tap-shifted kernel spread better.)
        ((f >= 128 && ceil_div(f, 256) * ceil_div(n * ((p.npix + 7) / 8 * 8), 256) * 16 >= rt->num_cu * 3) || variant == 5)) {
        const int st = launch_conv_pw_gemm(rt, dtype, x, w, bias, residual, y, n, c, p.npix, f, act);
        if (st >= 0)
            return st;
    }
    const bool pointwise_gemm = r == 1 && s == 1 && ph == 0 && pw == 0 && same_s1 && (p.npix % 8 == 0) && c % 64 == 0;
    // big-plane pointwise layers with >= 256 filters and channels are plain batched GEMMs (LDS-DMA kernels); everything else whose
    // output extent is ceil(input / stride) goes to the tap-shifted implicit GEMM of conv_s1.hip (measured per
    // ResNet-50 layer with tools/conv_bench.py)
    // a single K-step leaves nothing to pipeline: the small generic tile (more workgroups per CU) hides the latency better
    const bool one_kstep = (long)c * r * s <= 64 && f >= 128;
    // (an earlier rule sent 56x56 pointwise layers with >= 128 filters and <= 256 channels to the small generic tile; since the
    // LDS-staged epilogue conv_s1 wins there too: C256->F128 @56x56 93 vs 112 us)
    const bool big_plane_pointwise = false;
    // conv_pw_kernel (conv_s1.hip) candidates; IROCM_CONV_PW=2 (tuning hook) sends them there ahead of the two rules above
    static const int pw_pref = getenv("IROCM_CONV_PW") ? atoi(getenv("IROCM_CONV_PW")) : 1;
    const bool pw_shape = r == 1 && s == 1 && ph == 0 && pw == 0 && groups == 1 && c % 64 == 0 && c <= 256 && f > 64;
    // measured (tools/conv_bench.py): with <= 128 input channels conv_pw wins everywhere (C64->F256 @56x56 64 vs 89 us generic,
    // C128->F512 @28x28 44 vs 59 us conv_s1); at C = 256 its 100 KiB of LDS leaves one workgroup per CU and it loses
    const bool pw_default = pw_pref >= 1 && pw_shape && c <= 128 && p.npix % 2 == 0 && sh == 1 && sw == 1;
    if (variant < 0 && (one_kstep || big_plane_pointwise) && !(pw_default || (pw_pref == 2 && pw_shape)))
        goto generic;
    if (groups == 1 && variant != 1 && !(variant == 3 && pointwise_gemm) &&
        // batched-GEMM route by default only for long-K pointwise layers on big planes: with K <= 512 its 256^2 tiles run 8
        // K-tiles each and the per-tile prologue + epilogue dominates (C512->F256 @28x28: 88 us vs 66 us on conv_s1)
        (variant == 2 || variant == 4 || variant == 6 || variant == 7 || residual || !(pointwise_gemm && f >= 256 && c >= 1024 && p.npix >= 2048))) {
        rt->last_conv_route = "tap_shifted";
        const int st = launch_conv_s1(rt, dtype, x, w, bias, residual, y, (int)n, (int)c, (int)h, (int)wd, (int)f, (int)r, (int)s,
                                      ph, pw, sh, sw, dh, dw, p.oh, p.ow, act);
        if (st >= 0)
            return st;
    }
    // pointwise convolution == batched GEMM  Y[n] = W[F x C] . X[n][C x HW]  (A broadcast over batch)
    if (variant != 1 && !residual && r == 1 && s == 1 && ph == 0 && pw == 0 && sh == 1 && sw == 1 && groups == 1 && (p.npix % 8 == 0) &&
        c % 64 == 0) {
        rt->last_conv_route = "batched_gemm";
        return infini_rocm_matmul(rt, dtype, w, x, bias, y, n, f, p.npix, c, 0, 0, 0, (int64_t)c * p.npix,
                                  0, bias ? 1 : 0, 0, act);
    }
generic:
    p.tiles_m = (int)ceil_div(p.fg, 128);
    p.tiles_p = (int)ceil_div(p.npix, 128);
    const long blocks = (long)n * groups * p.tiles_m * p.tiles_p;
    IROCM_CHECK_ARG(blocks < (1l << 31), "conv2d: too many tiles");
    if (dtype == INFINI_DT_BF16)
        hipLaunchKernelGGL(conv_igemm16<Bf16Traits>, dim3((unsigned)blocks), dim3(256), 0, rt->stream, p);
    else
        hipLaunchKernelGGL(conv_igemm16<F16Traits>, dim3((unsigned)blocks), dim3(256), 0, rt->stream, p);
    IROCM_LAUNCH_CHECK("conv_igemm16");
    rt->last_conv_route = "generic";
    return INFINI_ROCM_OK;
}

int infini_rocm_conv2d_last_route(infiniRocmRuntime_t rt, const char **route) {
    IROCM_CHECK_ARG(rt && route, "NULL argument");
    *route = rt->last_conv_route;
    return INFINI_ROCM_OK;
}

int infini_rocm_conv2d_set_variant(infiniRocmRuntime_t rt, int variant) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(variant >= -1 && variant <= 7, "conv2d: bad variant %d", variant);
    rt->conv_variant = variant;
    return INFINI_ROCM_OK;
}

} // extern "C"
