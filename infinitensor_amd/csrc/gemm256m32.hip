// gemm256m32: the 256x256x64 staggered GEMM (see gemm256.hip) on v_mfma_f32_32x32x16_{bf16,f16}.
// Same tile / LDS-DMA / barrier schedule; only the fragment geometry changes: a wave's 128x64 sub-tile is
// 4 x 2 tiles of 32x32, an operand fragment is 16 bytes per lane = 8 consecutive k of row (lane & 31),
// k-half (lane >> 5); one K-tile is 4 k-steps of 16. The 32x32 shape has the higher per-instruction
// ceiling (MI355X_MICROARCH.md: 2382 vs 2075 TFLOP/s micro-benchmark) and half the instruction count.
//  * K-major image [256 rows][64 k]: physical 16-B chunk = (kk*2 + (lane>>5)) ^ ((row>>1) & 7); a
//    ds_read_b128 lane group {0-3,12-15,20-27} then covers 16 distinct 16-B slots of the 256-B bank row.
//  * M/N-major image [64 k][256 cols]: a half-wave of ds_read_b64_tr_b16 touches 4 k-rows x 64 bytes; the
//    32-B chunk index is XORed with 2*(k & 3) (16-B chunk index ^ ((k & 3) << 2)) => 8 distinct 32-B slots.
#include "gemm256_common.h"

namespace irocm {
namespace g256 {

struct Bf16M32 {
    __device__ static inline f32x16 mfma(s16x8_t a, s16x8_t b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
struct F16M32 {
    __device__ static inline f32x16 mfma(s16x8_t a, s16x8_t b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};

// M/N-major staging for the 32x32 geometry: 16-B chunk index ^ ((k & 3) << 2)
__device__ __forceinline__ void offs_mn32(unsigned (&off)[4], long ld, int col0, int cols, int w, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = w * 4 + i;
        const int kr = piece * 2 + (lane >> 5);
        const int c_log = (lane & 31) ^ ((kr & 3) << 2);
        int gc = col0 + c_log * 8;
        gc = gc <= cols - 8 ? gc : cols - 8;
        off[i] = (unsigned)(((long)kr * ld + gc) * 2);
    }
}

template <typename Tr, typename M, bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(512, 2) void gemm256m32_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 2, wc = w & 3;

    const unsigned per_batch = (unsigned)p.tiles_m * p.tiles_n;
    unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
    const int ib = wg / per_batch;
    wg -= ib * per_batch;
    constexpr int GROUP_M = 8;
    const unsigned per_group = GROUP_M * p.tiles_n;
    const unsigned group = wg / per_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int tm = first_m + (wg % per_group) % gsz;
    const int tn = (wg % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const unsigned short *A = (const unsigned short *)p.a + (long)ib * p.a_bs;
    const unsigned short *B = (const unsigned short *)p.b + (long)ib * p.b_bs;
    const long lda = A_KMAJOR ? p.a_rs : p.a_cs;
    const long ldb = B_KMAJOR ? p.b_cs : p.b_rs;
    const int nk = p.k / BK;

    unsigned a_off[4], b_off[4];
    if constexpr (A_KMAJOR) offs_k(a_off, lda, m0, p.m, w, lane);
    else offs_mn32(a_off, lda, m0, p.m, w, lane);
    if constexpr (B_KMAJOR) offs_k(b_off, ldb, n0, p.n, w, lane);
    else offs_mn32(b_off, ldb, n0, p.n, w, lane);
    const long a_step = A_KMAJOR ? (long)BK * 2 : (long)BK * lda * 2;
    const long b_step = B_KMAJOR ? (long)BK * 2 : (long)BK * ldb * 2;
    auto stage_a = [&](int buf, int kt) {
        stage4((const char *)A + (long)kt * a_step, a_off, smem + buf * BUF_BYTES, w);
    };
    auto stage_b = [&](int buf, int kt) {
        stage4((const char *)B + (long)kt * b_step, b_off, smem + buf * BUF_BYTES + OPER_BYTES, w);
    };

    // ---- per-lane LDS read addresses -------------------------------------------------------------
    const unsigned lds0 = (unsigned)(unsigned long)IROCM_LDS_PTR(smem);
    const int l31 = lane & 31, hi = lane >> 5;
    // K-major: row R0 + l31, chunk (kk*2 + hi) ^ ((l31 >> 1) & 7): kk XORs address bits 5..6
    const unsigned kmaj_lane = (unsigned)(l31 * 128 + ((hi ^ ((l31 >> 1) & 7)) & 7) * 16);
    unsigned a_k[4], b_k[4]; // [kk]
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        a_k[kk] = lds0 + wr * (128 * 128) + (kmaj_lane ^ (kk * 32));
        b_k[kk] = lds0 + OPER_BYTES + wc * (64 * 128) + (kmaj_lane ^ (kk * 32));
    }
    // M/N-major: 16-lane group g: col half g & 1, k half g >> 1; lane p supplies k-row (p >> 2), cols (p & 3) * 4
    const int p15 = lane & 15, g4 = lane >> 4;
    const int f = (p15 >> 2) & 3;
    const unsigned mn_lane = (unsigned)(((g4 >> 1) * 8 + (p15 >> 2)) * 512 + (g4 & 1) * 32 + ((p15 >> 1) & 1) * 16 + (p15 & 1) * 8);
    unsigned a_mn[4], b_mn[2]; // [tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
        a_mn[i] = lds0 + mn_lane + (((wr * 4 + i) ^ f)) * 64;
#pragma unroll
    for (int j = 0; j < 2; ++j)
        b_mn[j] = lds0 + OPER_BYTES + mn_lane + (((wc * 2 + j) ^ f)) * 64;
    auto flip_buf = [&](auto toc) {
        constexpr int d = decltype(toc)::value ? BUF_BYTES : -BUF_BYTES;
        if constexpr (A_KMAJOR) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) a_k[kk] += d;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) a_mn[i] += d;
        }
        if constexpr (B_KMAJOR) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) b_k[kk] += d;
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) b_mn[j] += d;
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = 0.f;

    using FA = Frag<A_KMAJOR>;
    using FB = Frag<B_KMAJOR>;
    FA aq[2][4];            // [m-tile in sub][kk]
    FB bq0[4], bq1[4];      // [kk]
    // A sub-tile q = m-tiles 2q, 2q+1; B sub-tile q = n-tile q
    auto read_a = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        sfor<2>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            sfor<4>([&](auto kc) {
                constexpr int kk = decltype(kc)::value;
                if constexpr (A_KMAJOR) {
                    aq[i][kk].v = lds_read_b128<(q * 2 + i) * 4096>(a_k[kk]);
                } else {
                    aq[i][kk].lo = lds_read_tr_b64<kk * 8192>(a_mn[q * 2 + i]);
                    aq[i][kk].hi = lds_read_tr_b64<kk * 8192 + 2048>(a_mn[q * 2 + i]);
                }
            });
        });
    };
    auto read_b = [&](auto qc, FB(&bq)[4]) {
        constexpr int q = decltype(qc)::value;
        sfor<4>([&](auto kc) {
            constexpr int kk = decltype(kc)::value;
            if constexpr (B_KMAJOR) {
                bq[kk].v = lds_read_b128<q * 4096>(b_k[kk]);
            } else {
                bq[kk].lo = lds_read_tr_b64<kk * 8192>(b_mn[q]);
                bq[kk].hi = lds_read_tr_b64<kk * 8192 + 2048>(b_mn[q]);
            }
        });
    };
    // 16 MFMAs: (A sub qa: 2 m-tiles) x (both n-tiles) x 4 k-steps, swapped operands; 4 independent
    // accumulators are interleaved so that a dependent MFMA is 4 issue slots away.
    auto compute = [&](auto qac) {
        constexpr int qa = decltype(qac)::value;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[qa * 2 + i][0] = M::mfma(bq0[kk].get(), aq[i][kk].get(), acc[qa * 2 + i][0]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[qa * 2 + i][1] = M::mfma(bq1[kk].get(), aq[i][kk].get(), acc[qa * 2 + i][1]);
        }
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    auto ktile = [&](auto bufc, int kt) {
        constexpr int buf = decltype(bufc)::value;
        using OB = std::integral_constant<int, buf ^ 1>;
        // L1
        read_b(I0{}, bq0);
        read_b(I1{}, bq1);
        read_a(I0{});
        if (kt + 1 < nk)
            stage_a(buf ^ 1, kt + 1);
        wait_lgkm0();
        barrier();
        // C1: 16 MFMAs
        __builtin_amdgcn_s_setprio(1);
        compute(I0{});
        __builtin_amdgcn_s_setprio(0);
        barrier();
        // L2
        read_a(I1{});
        flip_buf(OB{});
        if (kt + 2 < nk) {
            stage_b(buf, kt + 2);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        wait_lgkm0();
        barrier();
        // C2
        __builtin_amdgcn_s_setprio(1);
        compute(I1{});
        __builtin_amdgcn_s_setprio(0);
        barrier();
    };
    stage_b(0, 0);
    stage_a(0, 0);
    if (nk > 1) {
        stage_b(1, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    barrier();
    if (wr == 1)
        barrier();
    for (int kt = 0; kt < nk; kt += 2) {
        ktile(I0{}, kt);
        if (kt + 1 < nk)
            ktile(I1{}, kt + 1);
    }
    if (wr == 0)
        barrier();

    // ---- epilogue: lane holds row m = l31 of tile i, cols (r&3) + 8*(r>>2) + 4*hi of tile j -------------
    unsigned short *C = (unsigned short *)p.c + (long)ib * p.m * p.n;
    const unsigned short *bias = (const unsigned short *)p.bias;
    const bool interior = (m0 + BM <= p.m) && (n0 + BN <= p.n) && (p.n % 4 == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + wr * 128 + i * 32 + l31;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = n0 + wc * 64 + j * 32 + q * 8 + hi * 4;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    v[r] = acc[i][j][q * 4 + r];
                if (interior) {
                    if (bias) {
                        const unsigned short *bp = bias + (long)ib * p.bias_b + (long)row * p.bias_m + (long)col * p.bias_n;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            v[r] += Tr::to_f32(bp[(long)r * p.bias_n]);
                    }
                    if (p.act) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            v[r] = apply_act(v[r], p.act);
                    }
                    u32x2_t pk;
                    pk[0] = (unsigned)Tr::from_f32(v[0]) | ((unsigned)Tr::from_f32(v[1]) << 16);
                    pk[1] = (unsigned)Tr::from_f32(v[2]) | ((unsigned)Tr::from_f32(v[3]) << 16);
                    *(u32x2_t *)(C + c_off(p, row, col)) = pk;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (row < p.m && col + r < p.n) {
                            float x = v[r];
                            if (bias)
                                x += Tr::to_f32(bias[(long)ib * p.bias_b + (long)row * p.bias_m + (long)(col + r) * p.bias_n]);
                            C[c_off(p, row, col + r)] = Tr::from_f32(apply_act(x, p.act));
                        }
                    }
                }
            }
        }
    }
}

} // namespace g256

template <typename Tr, typename M> static int launch256m32(infiniRocmRuntime_t rt, GemmArgs p, bool akm, bool bkm) {
    p.tiles_m = (int)ceil_div(p.m, g256::BM);
    p.tiles_n = (int)ceil_div(p.n, g256::BN);
    const unsigned grid = (unsigned)p.tiles_m * p.tiles_n * p.batch;
#define IROCM_G256(AK, BK_)                                                                        \
    do {                                                                                           \
        auto kern = g256::gemm256m32_kernel<Tr, M, AK, BK_>;                                       \
        static bool attr_done = false;                                                             \
        if (!attr_done) {                                                                          \
            IROCM_HIP(hipFuncSetAttribute((const void *)kern,                                      \
                                          hipFuncAttributeMaxDynamicSharedMemorySize,              \
                                          g256::LDS_BYTES));                                       \
            attr_done = true;                                                                      \
        }                                                                                          \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), g256::LDS_BYTES, rt->stream, p);           \
    } while (0)
    if (akm && bkm) IROCM_G256(true, true);
    else if (akm && !bkm) IROCM_G256(true, false);
    else if (!akm && bkm) IROCM_G256(false, true);
    else IROCM_G256(false, false);
#undef IROCM_G256
    IROCM_LAUNCH_CHECK("gemm256m32");
    return INFINI_ROCM_OK;
}

int launch_gemm256m32(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, bool akm, bool bkm) {
    return dtype == INFINI_DT_BF16 ? launch256m32<Bf16Traits, g256::Bf16M32>(rt, p, akm, bkm)
                                   : launch256m32<F16Traits, g256::F16M32>(rt, p, akm, bkm);
}

} // namespace irocm
