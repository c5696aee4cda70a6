// gemm256w4: 256x256x64 tile with FOUR waves (2 x 2), each owning a 128 x 128 block of C — 8 x 8 MFMA 16x16x32
// tiles, 256 fp32 accumulators per lane (the unified 512-register file of a one-wave-per-SIMD kernel: accumulators
// in AGPRs, operands in VGPRs).
//
// Why: the 8-wave kernel (gemm256.hip, wave tile 128 x 64) reads 192 KiB of LDS per K-tile and the DMA writes 64 KiB;
// at 128 B/clk that is 2048 clk — exactly the MFMA time of the tile, so matrix pipe and LDS are co-limited (measured
// 62 % MFMA-busy). A 128 x 128 wave tile needs (128 + 128) x 64 x 2 B = 32 KiB per wave, 128 KiB per K-tile (+ 64 KiB
// DMA) = 1536 clk: 25 % head-room. With one wave per SIMD nothing else hides latency, so every wave software-pipelines
// itself: the K-tile is two k-steps of 32; while the 64 MFMAs of one k-step run out of one fragment register set,
// the ds_reads of the next k-step fill the other set, ONE read (or, in the second half, one LDS-DMA issue of tile
// t+2) slotted after each MFMA — an MFMA occupies the pipe for 16 clk but its issue only 4, the other issue slots
// are free (first version: 2-4 reads + 2 DMAs after every 8 MFMAs = 980 TF; the bursts stalled the pipe). One s_barrier per K-tile, between the halves: by then the wave has retired every LDS read of tile t
// (buffer free for tile t+2) and its own DMA pieces of tile t+1 have landed (vmcnt(0); issued a full tile earlier).
// LDS images, swizzles, DMA staging and the epilogue are those of gemm256.hip (gemm256_common.h).
//
// MEASURED (MI355X, bf16 4096^3 / 8192^3): 1064 / 1305 TFLOP/s vs 1331 / 1533 for the 8-wave staggered kernel on the
// same box — slower. The issue of one LDS-DMA instruction costs the issuing wave ~60-180 cycles (MI355X_MICROARCH.md,
// "LDS-DMA piece issue cost"), 16 of them per wave per K-tile, and a wave alone on its SIMD has no partner to hide
// that behind. Kept as matmul variant 5 for A/B runs; the heuristic never selects it.
#include "gemm256_common.h"

namespace irocm {
namespace g256 {

template <typename Tr, bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(256, 1) void gemm256w4_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 1, wc = w & 1;

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

    // staging: 32 one-KiB pieces per operand; this wave plays the two virtual staging waves 2w and 2w+1 of the
    // 8-wave layout (pieces 8w .. 8w+7)
    unsigned a_off[2][4], b_off[2][4];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        if constexpr (A_KMAJOR) offs_k(a_off[v], lda, m0, p.m, w * 2 + v, lane);
        else offs_mn(a_off[v], lda, m0, p.m, w * 2 + v, lane);
        if constexpr (B_KMAJOR) offs_k(b_off[v], ldb, n0, p.n, w * 2 + v, lane);
        else offs_mn(b_off[v], ldb, n0, p.n, w * 2 + v, lane);
    }
    const long a_step = A_KMAJOR ? (long)BK * 2 : (long)BK * lda * 2;
    const long b_step = B_KMAJOR ? (long)BK * 2 : (long)BK * ldb * 2;
    // one LDS-DMA instruction (1 KiB): q in [0, 16): operand q >> 3 (0 = A, 1 = B), virtual wave (q >> 2) & 1, piece q & 3
    auto dma_one = [&](int buf, int kt, auto qc) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value;
        constexpr int oper = q >> 3, v = (q >> 2) & 1, i = q & 3;
        const char *ubase = oper == 0 ? (const char *)A + (long)kt * a_step : (const char *)B + (long)kt * b_step;
        const unsigned off = oper == 0 ? a_off[v][i] : b_off[v][i];
        char *dst = smem + buf * BUF_BYTES + oper * OPER_BYTES + ((w * 2 + v) * 4 + i) * 1024;
        __builtin_amdgcn_global_load_lds(IROCM_GLB_PTR(ubase + (unsigned long)off), IROCM_LDS_PTR(dst), 16, 0, 0);
    };
    auto dma_tile = [&](int buf, int kt) __attribute__((always_inline)) { sfor<16>([&](auto qc) { dma_one(buf, kt, qc); }); };

    // ---- per-lane LDS read addresses ------------------------------------------------------------------
    const unsigned lds0 = (unsigned)(unsigned long)IROCM_LDS_PTR(smem);
    const int l15 = lane & 15, g4 = lane >> 4;
    const unsigned kmaj_lane = (unsigned)(l15 * 128 + (((g4 ^ (l15 >> 1)) & 3) | (((l15 >> 1) >> 2) << 2)) * 16);
    unsigned a_k[2], b_k[2];   // K-major: [ks]
    unsigned a_mn[8], b_mn[8]; // M/N-major: [tile]
    const int mnf = ((l15 >> 2) & 3) | ((g4 & 1) << 2);
    const unsigned mn_lane = (unsigned)((g4 * 8 + (l15 >> 2)) * 512 + (l15 & 1) * 8);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a_k[ks] = lds0 + wr * (128 * 128) + (kmaj_lane ^ (ks * 64));
        b_k[ks] = lds0 + OPER_BYTES + wc * (128 * 128) + (kmaj_lane ^ (ks * 64));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a_mn[i] = lds0 + mn_lane + ((((l15 >> 1) & 1) | (((wr * 8 + i) ^ mnf) << 1)) * 16);
        b_mn[i] = lds0 + OPER_BYTES + mn_lane + ((((l15 >> 1) & 1) | (((wc * 8 + i) ^ mnf) << 1)) * 16);
    }
    auto flip_buf = [&](auto toc) __attribute__((always_inline)) {
        constexpr int d = decltype(toc)::value ? BUF_BYTES : -BUF_BYTES;
        if constexpr (A_KMAJOR) { a_k[0] += d; a_k[1] += d; }
        else {
#pragma unroll
            for (int i = 0; i < 8; ++i) a_mn[i] += d;
        }
        if constexpr (B_KMAJOR) { b_k[0] += d; b_k[1] += d; }
        else {
#pragma unroll
            for (int i = 0; i < 8; ++i) b_mn[i] += d;
        }
    };

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    using FA = Frag<A_KMAJOR>;
    using FB = Frag<B_KMAJOR>;
    constexpr int NRA = A_KMAJOR ? 8 : 16, NRB = B_KMAJOR ? 8 : 16, NR = NRA + NRB; // LDS reads per k-step
    // read slot s of k-step KS into (fa, fb): B first (every MFMA row needs all of B), then A
    auto read_slot = [&](auto ksc, auto sc, FA(&fa)[8], FB(&fb)[8]) __attribute__((always_inline)) {
        constexpr int ks = decltype(ksc)::value, s = decltype(sc)::value;
        if constexpr (s < NRB) {
            if constexpr (B_KMAJOR) {
                fb[s].v = lds_read_b128<s * 2048>(b_k[ks]);
            } else {
                constexpr int j = s >> 1, hh = s & 1;
                if constexpr (hh == 0) fb[j].lo = lds_read_tr_b64<ks * 16384>(b_mn[j]);
                else fb[j].hi = lds_read_tr_b64<ks * 16384 + 2048>(b_mn[j]);
            }
        } else {
            constexpr int sa = s - NRB;
            if constexpr (A_KMAJOR) {
                fa[sa].v = lds_read_b128<sa * 2048>(a_k[ks]);
            } else {
                constexpr int i = sa >> 1, hh = sa & 1;
                if constexpr (hh == 0) fa[i].lo = lds_read_tr_b64<ks * 16384>(a_mn[i]);
                else fa[i].hi = lds_read_tr_b64<ks * 16384 + 2048>(a_mn[i]);
            }
        }
    };
    auto read_all = [&](auto ksc, FA(&fa)[8], FB(&fb)[8]) __attribute__((always_inline)) {
        sfor<NR>([&](auto sc) { read_slot(ksc, sc, fa, fb); });
    };
    // one half of a K-tile: 64 MFMAs out of (ca, cb). The issue slot after MFMA number idx carries one other
    // instruction: LDS read idx of k-step KS into (na, nb) for idx < NR [if READ], then the 16 LDS-DMA instructions of
    // tile kt_dma into buffer dma_buf [if DMA] — an MFMA holds the pipe for 16 clk, its issue takes 4.
    auto half = [&](FA(&ca)[8], FB(&cb)[8], auto ksc, FA(&na)[8], FB(&nb)[8], auto readc, auto dmac, int dma_buf,
                    int kt_dma) __attribute__((always_inline)) {
        constexpr bool READ = decltype(readc)::value, DMA = decltype(dmac)::value;
        sfor<64>([&](auto ic) {
            constexpr int idx = decltype(ic)::value, g = idx >> 3, j = idx & 7;
            acc[g][j] = Tr::mfma(cb[j].get(), ca[g].get(), acc[g][j]);
            fence_sched();
            if constexpr (READ && idx < NR)
                read_slot(ksc, std::integral_constant<int, (idx < NR ? idx : 0)>{}, na, nb);
            if constexpr (DMA && idx >= NR && idx < NR + 16)
                dma_one(dma_buf, kt_dma, std::integral_constant<int, (idx >= NR && idx < NR + 16 ? idx - NR : 0)>{});
            fence_sched();
        });
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using T = std::true_type;
    using F = std::false_type;
    FA ax[8], ay[8];
    FB bx[8], by[8];

    // Branch-free steady state: the last tiles prefetch a clamped (already consumed) tile and read fragments nobody
    // uses, instead of forking the loop body — with 256 live accumulators every control-flow merge costs spills.
    auto ktile = [&](auto bufc, int kt) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        using OB = std::integral_constant<int, buf ^ 1>;
        const int kt_dma = min(kt + 2, nk - 1);
        // first half: MFMAs of k-step 0 (x set) | reads of k-step 1 of this tile -> y set
        half(ax, bx, I1{}, ay, by, T{}, F{}, 0, 0);
        wait_lgkm0();
        flip_buf(OB{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // own pieces of tile kt+1 (issued a tile ago)
        barrier();                                        // everyone's; and everyone is done reading tile kt
        // second half: MFMAs of k-step 1 (y set) | reads of k-step 0 of tile kt+1 -> x set | DMA of tile kt+2 -> this buffer
        half(ay, by, I0{}, ax, bx, T{}, T{}, buf, kt_dma);
        wait_lgkm0();
    };

    dma_tile(0, 0);
    if (nk > 1)
        dma_tile(1, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier();
    read_all(I0{}, ax, bx);
    wait_lgkm0();
    for (int kt = 0; kt < nk; kt += 2) {
        ktile(I0{}, kt);
        if (kt + 1 < nk)
            ktile(I1{}, kt + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the clamped tail prefetches must land before the LDS is released

    // ---- epilogue ------------------------------------------------------------------------------------
    unsigned short *C = (unsigned short *)p.c + (long)ib * p.m * p.n;
    const unsigned short *bias = (const unsigned short *)p.bias;
    const bool interior = (m0 + BM <= p.m) && (n0 + BN <= p.n) && (p.n % 4 == 0);
    if (interior) {
        sfor<8>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int row = m0 + wr * 128 + i * 16 + l15;
            sfor<8>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int col = n0 + wc * 128 + j * 16 + g4 * 4;
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
                *(u32x2_t *)(C + c_off(p, row, col)) = pk;
            });
        });
    } else {
        sfor<8>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int row = m0 + wr * 128 + i * 16 + l15;
            sfor<8>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int col = n0 + wc * 128 + j * 16 + g4 * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (row < p.m && col + r < p.n) {
                        float x = acc[i][j][r];
                        if (bias)
                            x += Tr::to_f32(bias[(long)ib * p.bias_b + (long)row * p.bias_m + (long)(col + r) * p.bias_n]);
                        C[c_off(p, row, col + r)] = Tr::from_f32(apply_act(x, p.act));
                    }
                }
            });
        });
    }
}

} // namespace g256

int launch_gemm256(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, bool akm, bool bkm, int sched);

template <typename Tr> static int launch256w4(infiniRocmRuntime_t rt, GemmArgs p, bool akm, bool bkm) {
    p.tiles_m = (int)ceil_div(p.m, g256::BM);
    p.tiles_n = (int)ceil_div(p.n, g256::BN);
    const unsigned grid = (unsigned)p.tiles_m * p.tiles_n * p.batch;
#define IROCM_G256W4(AK, BK_)                                                                      \
    do {                                                                                           \
        auto kern = g256::gemm256w4_kernel<Tr, AK, BK_>;                                           \
        static bool attr_done = false;                                                             \
        if (!attr_done) {                                                                          \
            IROCM_HIP(hipFuncSetAttribute((const void *)kern,                                      \
                                          hipFuncAttributeMaxDynamicSharedMemorySize,              \
                                          g256::LDS_BYTES));                                       \
            attr_done = true;                                                                      \
        }                                                                                          \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), g256::LDS_BYTES, rt->stream, p);           \
    } while (0)
    // experiment kept for A/B runs: only the two A-K-major layouts (ONNX NN / NT) are instantiated
    if (akm && bkm) IROCM_G256W4(true, true);
    else if (akm && !bkm) IROCM_G256W4(true, false);
    else return launch_gemm256(rt, Tr::kDType, p, akm, bkm, 0);
#undef IROCM_G256W4
    IROCM_LAUNCH_CHECK("gemm256w4");
    return INFINI_ROCM_OK;
}

int launch_gemm256w4(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, bool akm, bool bkm) {
    return dtype == INFINI_DT_BF16 ? launch256w4<Bf16Traits>(rt, p, akm, bkm) : launch256w4<F16Traits>(rt, p, akm, bkm);
}

} // namespace irocm
