// gemm256: the headline GEMM kernel for gfx950 — 256x256x64 tile, 8 waves, 128 KiB LDS, one
// workgroup per CU (4096^3 = exactly 256 tiles = one wave of workgroups on the 256 CUs).
//
// Structure (derived from the bank / pipe rules of MI355X_MICROARCH.md, not from any library):
//  * waves: 2 (M) x 4 (N); wave (wr, wc) owns C rows wr*128..+127, cols wc*64..+63 = 8 x 4 MFMA
//    16x16 tiles, 128 fp32 accumulators per lane.
//  * LDS: two K-tile buffers, each [A 32 KiB | B 32 KiB]; filled by LDS-DMA (global_load_lds_dwordx4,
//    1 KiB per wave-instruction). K-major operands: image [256 rows][64 k], 16-byte chunk index XORed
//    with (row >> 1) & 7 (ds_read_b128 conflict-free); M/N-major operands: image [64 k][256 cols],
//    32-byte chunk index XORed with f(k) and read with ds_read_b64_tr_b16 (transpose read). The XOR is
//    applied on the global SOURCE address because the DMA destination is lane-linear.
//  * schedule per K-tile: 4 phases, each a LOAD segment (inline-asm ds_reads of one operand sub-tile,
//    counted waits) and a COMPUTE segment (16 MFMAs = one 64x32 C quadrant x K=64), separated by
//    s_barrier. The two wave rows run the same stream offset by ONE barrier interval, and waves w and
//    w+4 share a SIMD: in every interval each SIMD has one wave issuing MFMAs (s_setprio 1) while its
//    partner issues LDS reads / DMA — the matrix pipe never waits for an LDS read of its own wave.
//  * prefetch distance one full K-tile: tile t+2 is DMA'd into the buffer of tile t as soon as that
//    buffer's last reads have retired (B half after phase 2, A half after phase 3); the only vmcnt wait
//    (counted, never 0 in steady state) sits in phase 4 and guards tile t+1.
//  * ds_reads are inline asm on purpose: hipcc drains vmcnt(0) before any LDS read it can see while an
//    LDS-DMA is in flight (no alias info on the DMA), which would serialise the pipeline.
//
// WAR / RAW argument (intervals of group G0; G1 = G0 + 1):
//   reads of buffer b:  B sub-tiles in L1,L2 (intervals 0,2 / 1,3), A sub-tiles in L1,L3 (0,4 / 1,5);
//   every LOAD segment ends with lgkmcnt(0) BEFORE its barrier, so reads retire inside their interval.
//   DMA into buffer b:  B(t+2) issued in L3 (interval 4 / 5 >= 4 > 3), A(t+2) in L4 (6 / 7 > 5)  => WAR safe.
//   tile t+1 (other buffer) was issued in L3/L4 of tile t-1; each wave waits vmcnt(8) in L4(t)
//   (interval 6 / 7) and the barrier ending interval 7 precedes the first read at interval 8      => RAW safe.
#include "gemm_common.h"
#include <type_traits>
#include <utility>

namespace irocm {
namespace g256 {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int OPER_BYTES = 256 * 64 * 2; // 32 KiB per operand tile
constexpr int BUF_BYTES = 2 * OPER_BYTES; // A | B
constexpr int LDS_BYTES = 2 * BUF_BYTES;  // 128 KiB

template <typename F, int... I> __device__ __forceinline__ void sfor_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void sfor(F &&f) {
    sfor_impl(f, std::make_integer_sequence<int, N>{});
}

template <int OFF> __device__ __forceinline__ s16x8_t lds_read_b128(unsigned addr) {
    s16x8_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
template <int OFF> __device__ __forceinline__ s16x4_t lds_read_tr_b64(unsigned addr) {
    s16x4_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
// One MFMA operand fragment (8 x 16-bit). K-major: one ds_read_b128. M/N-major: two transpose reads
// whose halves are only joined into one 128-bit value AFTER the lgkmcnt wait (a v_mov issued between the
// asm read and the wait would copy stale registers: hipcc does not know the asm is a load).
template <bool KMAJOR> struct Frag;
template <> struct Frag<true> {
    s16x8_t v;
    __device__ __forceinline__ s16x8_t get() const { return v; }
};
template <> struct Frag<false> {
    s16x4_t lo, hi;
    __device__ __forceinline__ s16x8_t get() const {
        return s16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
};

__device__ __forceinline__ void wait_lgkm0() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// ---- staging (LDS-DMA) -----------------------------------------------------------------------
// K-major operand: 32 pieces of 8 rows; wave w issues pieces w*4 .. w*4+3.
__device__ __forceinline__ void stage_k(const unsigned short *base, long ld, int row0, int rows, int k0,
                                        char *lds_oper, int w, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = w * 4 + i;
        const int r = piece * 8 + (lane >> 3);
        const int c_log = (lane & 7) ^ ((r >> 1) & 7);
        int gr = row0 + r;
        gr = gr < rows ? gr : rows - 1;
        const unsigned short *src = base + (long)gr * ld + k0 + c_log * 8;
        __builtin_amdgcn_global_load_lds(IROCM_GLB_PTR(src), IROCM_LDS_PTR(lds_oper + piece * 1024), 16, 0, 0);
    }
}
// M/N-major operand: image [64 k][256 cols] (512-B rows), 32 pieces of 2 k-rows.
__device__ __forceinline__ int mn_f(int kr) { return (kr & 3) | (((kr >> 3) & 1) << 2); }
__device__ __forceinline__ void stage_mn(const unsigned short *base, long ld, int col0, int cols, int k0,
                                         char *lds_oper, int w, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = w * 4 + i;
        const int kr = piece * 2 + (lane >> 5);
        const int c_log = (lane & 31) ^ (mn_f(kr) << 1);
        int gc = col0 + c_log * 8;
        gc = gc <= cols - 8 ? gc : cols - 8;
        const unsigned short *src = base + (long)(k0 + kr) * ld + gc;
        __builtin_amdgcn_global_load_lds(IROCM_GLB_PTR(src), IROCM_LDS_PTR(lds_oper + piece * 1024), 16, 0, 0);
    }
}

template <typename Tr, bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmArgs p) {
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

    auto stage_a = [&](int buf, int kt) {
        char *dst = smem + buf * BUF_BYTES;
        if constexpr (A_KMAJOR) stage_k(A, lda, m0, p.m, kt * BK, dst, w, lane);
        else stage_mn(A, lda, m0, p.m, kt * BK, dst, w, lane);
    };
    auto stage_b = [&](int buf, int kt) {
        char *dst = smem + buf * BUF_BYTES + OPER_BYTES;
        if constexpr (B_KMAJOR) stage_k(B, ldb, n0, p.n, kt * BK, dst, w, lane);
        else stage_mn(B, ldb, n0, p.n, kt * BK, dst, w, lane);
    };

    // ---- per-lane LDS read addresses (byte offsets in the workgroup's LDS) ----------------------
    const unsigned lds0 = (unsigned)(unsigned long)IROCM_LDS_PTR(smem);
    const int l15 = lane & 15, g4 = lane >> 4;
    // K-major fragment of row r0 + l15: chunk (ks*4 + g4) ^ (l15 >> 1); ks flips address bit 6.
    const unsigned kmaj_lane = (unsigned)(l15 * 128 + (((g4 ^ (l15 >> 1)) & 3) | (((l15 >> 1) >> 2) << 2)) * 16);
    unsigned a_k[2][2], b_k[2][2];   // [buf][ks]
    unsigned a_mn[2][8], b_mn[2][4]; // [buf][tile]
    // M/N-major fragment: lane p = l15 supplies k-row (p >> 2) (+ hh*4 + g4*8 + ks*32), 4 cols (p & 3)*4
    const int mnf = ((l15 >> 2) & 3) | ((g4 & 1) << 2);
    const unsigned mn_lane = (unsigned)((g4 * 8 + (l15 >> 2)) * 512 + (l15 & 1) * 8);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            a_k[b][ks] = lds0 + b * BUF_BYTES + wr * (128 * 128) + (kmaj_lane ^ (ks * 64));
            b_k[b][ks] = lds0 + b * BUF_BYTES + OPER_BYTES + wc * (64 * 128) + (kmaj_lane ^ (ks * 64));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c16 = ((l15 >> 1) & 1) | ((((wr * 8 + i) ^ mnf)) << 1);
            a_mn[b][i] = lds0 + b * BUF_BYTES + mn_lane + c16 * 16;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c16 = ((l15 >> 1) & 1) | ((((wc * 4 + j) ^ mnf)) << 1);
            b_mn[b][j] = lds0 + b * BUF_BYTES + OPER_BYTES + mn_lane + c16 * 16;
        }
    }

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Frag<A_KMAJOR> aq[4][2];
    Frag<B_KMAJOR> bq0[2][2], bq1[2][2];

    // read A sub-tile q (4 m-tiles x 2 k-steps) of buffer `buf` into aq
    auto read_a = [&](auto bufc, auto qc) {
        constexpr int buf = decltype(bufc)::value, q = decltype(qc)::value;
        sfor<4>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            sfor<2>([&](auto kc) {
                constexpr int ks = decltype(kc)::value;
                if constexpr (A_KMAJOR) {
                    aq[i][ks].v = lds_read_b128<(q * 4 + i) * 2048>(a_k[buf][ks]);
                } else {
                    aq[i][ks].lo = lds_read_tr_b64<ks * 16384>(a_mn[buf][q * 4 + i]);
                    aq[i][ks].hi = lds_read_tr_b64<ks * 16384 + 2048>(a_mn[buf][q * 4 + i]);
                }
            });
        });
    };
    auto read_b = [&](auto bufc, auto qc, Frag<B_KMAJOR>(&bq)[2][2]) {
        constexpr int buf = decltype(bufc)::value, q = decltype(qc)::value;
        sfor<2>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            sfor<2>([&](auto kc) {
                constexpr int ks = decltype(kc)::value;
                if constexpr (B_KMAJOR) {
                    bq[j][ks].v = lds_read_b128<(q * 2 + j) * 2048>(b_k[buf][ks]);
                } else {
                    bq[j][ks].lo = lds_read_tr_b64<ks * 16384>(b_mn[buf][q * 2 + j]);
                    bq[j][ks].hi = lds_read_tr_b64<ks * 16384 + 2048>(b_mn[buf][q * 2 + j]);
                }
            });
        });
    };
    // 16 MFMAs: C quadrant (A sub qa, B sub qb) x K = 64. Swapped operands: a lane ends up holding 4
    // consecutive n of one m row.
    auto compute = [&](auto qac, auto qbc, Frag<B_KMAJOR>(&bq)[2][2]) {
        constexpr int qa = decltype(qac)::value, qb = decltype(qbc)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[qa * 4 + i][qb * 2 + j] = Tr::mfma(bq[j][ks].get(), aq[i][ks].get(), acc[qa * 4 + i][qb * 2 + j]);
        __builtin_amdgcn_s_setprio(0);
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    auto ktile = [&](auto bufc, int kt) {
        constexpr int buf = decltype(bufc)::value;
        const bool pf = kt + 2 < nk; // prefetch tile kt+2 into THIS buffer
        // L1 | C1
        read_b(bufc, I0{}, bq0);
        read_a(bufc, I0{});
        wait_lgkm0();
        barrier();
        compute(I0{}, I0{}, bq0);
        barrier();
        // L2 | C2
        read_b(bufc, I1{}, bq1);
        wait_lgkm0();
        barrier();
        compute(I0{}, I1{}, bq1);
        barrier();
        // L3 | C3   (B half of this buffer is dead: start refilling it)
        read_a(bufc, I1{});
        if (pf)
            stage_b(buf, kt + 2);
        wait_lgkm0();
        barrier();
        compute(I1{}, I1{}, bq1);
        barrier();
        // L4 | C4   (A half dead too); guard tile kt+1 with a COUNTED wait
        if (pf) {
            stage_a(buf, kt + 2);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        barrier();
        compute(I1{}, I0{}, bq0);
        barrier();
    };

    // ---- prologue ------------------------------------------------------------------------------
    stage_b(0, 0);
    stage_a(0, 0);
    if (nk > 1) {
        stage_b(1, 1);
        stage_a(1, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    barrier();
    if (wr == 1)
        barrier(); // stagger: wave row 1 runs one barrier interval behind wave row 0

    for (int kt = 0; kt < nk; kt += 2) {
        ktile(I0{}, kt);
        if (kt + 1 < nk)
            ktile(I1{}, kt + 1);
    }
    if (wr == 0)
        barrier(); // balance the stagger

    // ---- epilogue ------------------------------------------------------------------------------
    unsigned short *C = (unsigned short *)p.c + (long)ib * p.m * p.n;
    const unsigned short *bias = (const unsigned short *)p.bias;
    const bool interior = (m0 + BM <= p.m) && (n0 + BN <= p.n) && (p.n % 4 == 0);
    if (interior) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = m0 + wr * 128 + i * 16 + l15;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + wc * 64 + j * 16 + g4 * 4;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    v[r] = acc[i][j][r];
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
                *(u32x2_t *)(C + (long)row * p.n + col) = pk;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = m0 + wr * 128 + i * 16 + l15;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + wc * 64 + j * 16 + g4 * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (row < p.m && col + r < p.n) {
                        float v = acc[i][j][r];
                        if (bias)
                            v += Tr::to_f32(bias[(long)ib * p.bias_b + (long)row * p.bias_m + (long)(col + r) * p.bias_n]);
                        C[(long)row * p.n + col + r] = Tr::from_f32(apply_act(v, p.act));
                    }
                }
            }
        }
    }
}

} // namespace g256

static bool al16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

bool gemm256_supported(const GemmArgs &p, bool akm, bool bkm) {
    if (p.k % g256::BK != 0 || p.k < g256::BK)
        return false;
    if (!al16(p.a) || !al16(p.b) || (p.a_bs % 8) || (p.b_bs % 8))
        return false;
    if (!akm && (p.m % 8 != 0 || p.m < 8))
        return false;
    if (!bkm && (p.n % 8 != 0 || p.n < 8))
        return false;
    if ((((uintptr_t)p.c) & 7) != 0)
        return false;
    return true;
}

template <typename Tr> static int launch256(infiniRocmRuntime_t rt, GemmArgs p, bool akm, bool bkm) {
    p.tiles_m = (int)ceil_div(p.m, g256::BM);
    p.tiles_n = (int)ceil_div(p.n, g256::BN);
    const unsigned grid = (unsigned)p.tiles_m * p.tiles_n * p.batch;
#define IROCM_G256(AK, BK_)                                                                        \
    do {                                                                                           \
        auto kern = g256::gemm256_kernel<Tr, AK, BK_>;                                             \
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
    IROCM_LAUNCH_CHECK("gemm256");
    return INFINI_ROCM_OK;
}

int launch_gemm256(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, bool akm, bool bkm) {
    return dtype == INFINI_DT_BF16 ? launch256<Bf16Traits>(rt, p, akm, bkm) : launch256<F16Traits>(rt, p, akm, bkm);
}

} // namespace irocm
