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
    int act;
    int tiles_m, tiles_n;
};

// 16-bit element traits: how to feed v_mfma_f32_16x16x32_{bf16,f16} and convert on store.
struct Bf16Traits {
    static constexpr int kDType = INFINI_DT_BF16;
    __device__ static inline f32x4 mfma(s16x8_t a, s16x8_t b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                       __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
    // round-to-nearest-even, NaN preserved (same rounding torch / the oracle use)
    __device__ static inline unsigned short from_f32(float f) {
        unsigned int u = __builtin_bit_cast(unsigned int, f);
        if ((u & 0x7fffffffu) > 0x7f800000u)
            return (unsigned short)((u >> 16) | 0x40);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (unsigned short)(u >> 16);
    }
    __device__ static inline float to_f32(unsigned short h) {
        return __builtin_bit_cast(float, ((unsigned int)h) << 16);
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
};

// Fused epilogue activation (reference ActType: include/core/common.h — None/Relu/Sigmoid/Tanh).
__device__ static inline float apply_act(float v, int act) {
    switch (act) {
    case 1: return v > 0.f ? v : 0.f;
    case 2: return 1.f / (1.f + __expf(-v));
    case 3: return tanhf(v);
    default: return v;
    }
}

// Bijective XCD-aware remap of a linear workgroup id: consecutive ids returned to one XCD
// (dispatch is round-robin over the 8 XCDs: block b runs on XCD b % 8 — speed only).
__device__ static inline unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg / kNumXcd, r = nwg % kNumXcd;
    const unsigned xcd = bid % kNumXcd, idx = bid / kNumXcd;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

} // namespace irocm
