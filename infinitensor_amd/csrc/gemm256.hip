// gemm256: the headline GEMM kernels for gfx950 — 256x256x64 tile, 8 waves, 128 KiB LDS, one
// workgroup per CU (4096^3 = exactly 256 tiles = one wave of workgroups on the 256 CUs).
//
// Common structure (derived from the bank / pipe rules of MI355X_MICROARCH.md, not from any library):
//  * waves: 2 (M) x 4 (N); wave (wr, wc) owns C rows wr*128..+127, cols wc*64..+63 = 8 x 4 MFMA
//    16x16 tiles, 128 fp32 accumulators per lane (swapped MFMA operands: a lane holds 4 consecutive n).
//  * LDS: two K-tile buffers, each [A 32 KiB | B 32 KiB], filled by LDS-DMA (global_load_lds_dwordx4,
//    1 KiB per wave-instruction; per-lane 32-bit source offsets are computed once per workgroup, a
//    K-tile step only advances a wave-uniform base). K-major operands: image [256 rows][64 k], 16-byte
//    chunk index XORed with (row >> 1) & 7 (ds_read_b128 conflict-free); M/N-major operands: image
//    [64 k][256 cols], 32-byte chunk index XORed with f(k), read with ds_read_b64_tr_b16 (transpose
//    read). The XOR is applied on the global SOURCE address because the DMA destination is lane-linear.
//    rocprofv3: SQ_LDS_BANK_CONFLICT = 0 for all four layouts (profiles/r01_gemm256_v1_pmc.json).
//  * ds_reads are inline asm on purpose: hipcc drains vmcnt(0) before any LDS read it can see while an
//    LDS-DMA is in flight (no alias info on the DMA), which would serialise the pipeline.
//
// Schedule ("staggered"): per K-tile 2 phases per wave, LOAD (ds_reads + DMA issue) | COMPUTE (32 MFMAs),
//    4 s_barriers per K-tile; the two wave rows run offset by one barrier interval and waves w / w+4 share
//    a SIMD, so each SIMD always has one wave in COMPUTE. (The first version used 16-MFMA segments and
//    8 barriers: 57 % MFMA-busy — each barrier + restart costs ~150 cycles of matrix-pipe idle.)
//      WAR/RAW (intervals of G0; G1 = +1): reads of tile t: B0,B1,A0 in L1 (0/1), A1 in L2 (2/3), each
//      LOAD ends with lgkmcnt(0) before its barrier. DMA A(t+1) -> other buffer in L1(t) (that A half
//      was last read in L2(t-1) = -2/-1); DMA B(t+2) -> this buffer in L2(t) (B half last read in L1(t)
//      = 0/1). Tile t+1 = {B issued L2(t-1), A issued L1(t)}; vmcnt(4) in L2(t) + the barrier ending
//      interval 3 precede its first read at interval 4.
// Alternatives that were built, measured slower and removed again (numbers in DESIGN.md's tuning log): a per-wave
// software-pipelined schedule with one barrier per K-tile (spills), a v_mfma_f32_32x32x16 variant, a 4-wave
// 128x128-wave-tile variant with AGPR accumulators, an 8-byte-store epilogue.
// This one-shot kernel serves grids of at most one tile per CU and split-K; gemm256p_kernel.h is its persistent
// multi-tile edition (same inner loop).
#include "gemm256_common.h"

namespace irocm {
namespace g256 {

template <typename Tr, bool A_KMAJOR, bool B_KMAJOR, bool SPLITK = false>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 2, wc = w & 3;

    const unsigned per_batch = (unsigned)p.tiles_m * p.tiles_n;
    unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
    int sp = 0; // K slice of this workgroup (slices of one tile are neighbours: they share nothing but the output tile)
    if constexpr (SPLITK) {
        sp = wg % p.splitk;
        wg /= p.splitk;
    }
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
    int nk = p.k / BK;
    int kt0 = 0; // first K-tile of this workgroup
    if constexpr (SPLITK) {
        const int per = (nk + p.splitk - 1) / p.splitk;
        kt0 = sp * per;
        nk = min(per, nk - kt0); // host guarantees >= 1
    }

    unsigned a_off[4], b_off[4];
    if constexpr (A_KMAJOR) offs_k(a_off, lda, m0, p.m, w, lane);
    else offs_mn(a_off, lda, m0, p.m, w, lane);
    if constexpr (B_KMAJOR) offs_k(b_off, ldb, n0, p.n, w, lane);
    else offs_mn(b_off, ldb, n0, p.n, w, lane);
    const long a_step = A_KMAJOR ? (long)BK * 2 : (long)BK * lda * 2; // bytes per K-tile (wave-uniform)
    const long b_step = B_KMAJOR ? (long)BK * 2 : (long)BK * ldb * 2;
    auto stage_a = [&](int buf, int kt) {
        stage4((const char *)A + (long)(kt0 + kt) * a_step, a_off, smem + buf * BUF_BYTES, w);
    };
    auto stage_b = [&](int buf, int kt) {
        stage4((const char *)B + (long)(kt0 + kt) * b_step, b_off, smem + buf * BUF_BYTES + OPER_BYTES, w);
    };

    // ---- per-lane LDS read addresses (byte offsets in the workgroup's LDS) ----------------------
    const unsigned lds0 = (unsigned)(unsigned long)IROCM_LDS_PTR(smem);
    const int l15 = lane & 15, g4 = lane >> 4;
    // K-major fragment of row r0 + l15: chunk (ks*4 + g4) ^ (l15 >> 1); ks flips address bit 6.
    const unsigned kmaj_lane = (unsigned)(l15 * 128 + (((g4 ^ (l15 >> 1)) & 3) | (((l15 >> 1) >> 2) << 2)) * 16);
    // Addresses of the CURRENT read buffer; flip_buf() moves them to the other K-tile buffer (one VALU
    // add per address register per tile — cheaper than a second register set, and registers are the
    // scarce resource here: 128 accumulators + 96 operand registers per lane).
    unsigned a_k[2], b_k[2];   // K-major: [ks]
    unsigned a_mn[8], b_mn[4]; // M/N-major: [tile]
    // M/N-major fragment: lane p = l15 supplies k-row (p >> 2) (+ hh*4 + g4*8 + ks*32), 4 cols (p & 3)*4
    const int mnf = ((l15 >> 2) & 3) | ((g4 & 1) << 2);
    const unsigned mn_lane = (unsigned)((g4 * 8 + (l15 >> 2)) * 512 + (l15 & 1) * 8);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a_k[ks] = lds0 + wr * (128 * 128) + (kmaj_lane ^ (ks * 64));
        b_k[ks] = lds0 + OPER_BYTES + wc * (64 * 128) + (kmaj_lane ^ (ks * 64));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c16 = ((l15 >> 1) & 1) | ((((wr * 8 + i) ^ mnf)) << 1);
        a_mn[i] = lds0 + mn_lane + c16 * 16;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c16 = ((l15 >> 1) & 1) | ((((wc * 4 + j) ^ mnf)) << 1);
        b_mn[j] = lds0 + OPER_BYTES + mn_lane + c16 * 16;
    }
    auto flip_buf = [&](auto toc) { // toc = index of the buffer to read from next
        constexpr int d = decltype(toc)::value ? BUF_BYTES : -BUF_BYTES;
        if constexpr (A_KMAJOR) { a_k[0] += d; a_k[1] += d; }
        else {
#pragma unroll
            for (int i = 0; i < 8; ++i) a_mn[i] += d;
        }
        if constexpr (B_KMAJOR) { b_k[0] += d; b_k[1] += d; }
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) b_mn[j] += d;
        }
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    using FA = Frag<A_KMAJOR>;
    using FB = Frag<B_KMAJOR>;
    // read A sub-tile q (4 m-tiles x 2 k-steps) of buffer `buf`
    auto read_a = [&](auto qc, FA(&aq)[4][2]) {
        constexpr int q = decltype(qc)::value;
        sfor<4>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            sfor<2>([&](auto kc) {
                constexpr int ks = decltype(kc)::value;
                if constexpr (A_KMAJOR) {
                    aq[i][ks].v = lds_read_b128<(q * 4 + i) * 2048>(a_k[ks]);
                } else {
                    aq[i][ks].lo = lds_read_tr_b64<ks * 16384>(a_mn[q * 4 + i]);
                    aq[i][ks].hi = lds_read_tr_b64<ks * 16384 + 2048>(a_mn[q * 4 + i]);
                }
            });
        });
    };
    auto read_b = [&](auto qc, FB(&bq)[2][2]) {
        constexpr int q = decltype(qc)::value;
        sfor<2>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            sfor<2>([&](auto kc) {
                constexpr int ks = decltype(kc)::value;
                if constexpr (B_KMAJOR) {
                    bq[j][ks].v = lds_read_b128<(q * 2 + j) * 2048>(b_k[ks]);
                } else {
                    bq[j][ks].lo = lds_read_tr_b64<ks * 16384>(b_mn[q * 2 + j]);
                    bq[j][ks].hi = lds_read_tr_b64<ks * 16384 + 2048>(b_mn[q * 2 + j]);
                }
            });
        });
    };
    // 16 MFMAs: C quadrant (A sub qa, B sub qb) x K = 64
    auto compute = [&](auto qac, auto qbc, FA(&aq)[4][2], FB(&bq)[2][2]) {
        constexpr int qa = decltype(qac)::value, qb = decltype(qbc)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[qa * 4 + i][qb * 2 + j] =
                        Tr::mfma(bq[j][ks].get(), aq[i][ks].get(), acc[qa * 4 + i][qb * 2 + j]);
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // ======================= staggered LOAD | COMPUTE schedule ==================================
    FA aq[4][2];
    FB bq0[2][2], bq1[2][2];
    auto ktile = [&](auto bufc, int kt) {
        constexpr int buf = decltype(bufc)::value;
        using OB = std::integral_constant<int, buf ^ 1>;
        // L1
        read_b(I0{}, bq0);
        read_b(I1{}, bq1);
        read_a(I0{}, aq);
        if (kt + 1 < nk)
            stage_a(buf ^ 1, kt + 1);
        wait_lgkm0();
        barrier();
        // C1
        __builtin_amdgcn_s_setprio(1);
        compute(I0{}, I0{}, aq, bq0);
        compute(I0{}, I1{}, aq, bq1);
        __builtin_amdgcn_s_setprio(0);
        barrier();
        // L2
        read_a(I1{}, aq);
        flip_buf(OB{}); // every read of this tile is issued: next reads come from the other buffer
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
        compute(I1{}, I1{}, aq, bq1);
        compute(I1{}, I0{}, aq, bq0);
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
        barrier(); // stagger: wave row 1 runs one barrier interval behind wave row 0
    for (int kt = 0; kt < nk; kt += 2) {
        ktile(I0{}, kt);
        if (kt + 1 < nk)
            ktile(I1{}, kt + 1);
    }
    if (wr == 0)
        barrier(); // balance the stagger

    // ---- epilogue ------------------------------------------------------------------------------
    if constexpr (SPLITK) { // raw fp32 partial sums of this K slice; bias / activation / rounding in splitk_reduce
        float *P = p.partial + ((long)sp * p.batch + ib) * p.m * p.n;
        const bool inner = (m0 + BM <= p.m) && (n0 + BN <= p.n) && (p.n % 4 == 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = m0 + wr * 128 + i * 16 + l15;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + wc * 64 + j * 16 + g4 * 4;
                if (inner) {
                    *(f32x4 *)(P + (long)row * p.n + col) = acc[i][j];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (row < p.m && col + r < p.n)
                            P[(long)row * p.n + col + r] = acc[i][j][r];
                }
            }
        }
        return;
    }
    unsigned short *C = (unsigned short *)p.c + (long)ib * p.c_bs;
    const unsigned short *bias = (const unsigned short *)p.bias;
    const bool interior = (m0 + BM <= p.m) && (n0 + BN <= p.n) && (p.n % 4 == 0);
    if (interior && p.epi16 && (p.n % 8 == 0) && ((((uintptr_t)p.c) & 15) == 0)) {
        // 16-byte stores: lanes g4 / g4^1 swap half of a pair of column tiles, so each lane ends up with 8 consecutive
        // columns of one tile (the store tail of a one-tile-per-CU GEMM is issue-bound: half as many stores, twice as
        // wide: 4096^3 bf16 1246 -> 1275 TFLOP/s sustained, 1305 -> 1374 in short runs)
        const bool odd = g4 & 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = m0 + wr * 128 + i * 16 + l15;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                unsigned pk[2][2];
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    const int col = n0 + wc * 64 + (jp * 2 + t2) * 16 + g4 * 4;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        v[r] = acc[i][jp * 2 + t2][r];
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
                    pk[t2][0] = (unsigned)Tr::from_f32(v[0]) | ((unsigned)Tr::from_f32(v[1]) << 16);
                    pk[t2][1] = (unsigned)Tr::from_f32(v[2]) | ((unsigned)Tr::from_f32(v[3]) << 16);
                }
                // even lanes keep tile 2jp and receive its next 4 columns; odd lanes keep tile 2jp+1 and receive its previous 4
                const unsigned s0 = odd ? pk[0][0] : pk[1][0], s1 = odd ? pk[0][1] : pk[1][1];
                const unsigned r0 = (unsigned)__shfl_xor((int)s0, 16), r1 = (unsigned)__shfl_xor((int)s1, 16);
                u32x4_t o;
                if (odd) { o[0] = r0; o[1] = r1; o[2] = pk[1][0]; o[3] = pk[1][1]; }
                else { o[0] = pk[0][0]; o[1] = pk[0][1]; o[2] = r0; o[3] = r1; }
                const int col = n0 + wc * 64 + (jp * 2 + (odd ? 1 : 0)) * 16 + (g4 & ~1) * 4;
                *(u32x4_t *)(C + c_off(p, row, col)) = o;
            }
        }
    } else if (interior) {
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
                *(u32x2_t *)(C + c_off(p, row, col)) = pk;
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
                        C[c_off(p, row, col + r)] = Tr::from_f32(apply_act(v, p.act));
                    }
                }
            }
        }
    }
}

// out = round(act(sum_s partial[s] + bias)); one thread per 4 consecutive columns when n % 4 == 0
template <typename Tr>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs p) {
    const long mn = (long)p.m * p.n, total = (long)p.batch * mn;
    const long plane = total; // elements per split plane
    const unsigned short *bias = (const unsigned short *)p.bias;
    unsigned short *C = (unsigned short *)p.c;
    const bool vec = (p.n % 4 == 0);
    const long items = vec ? total / 4 : total;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const long e0 = vec ? it * 4 : it;
        const int cnt = vec ? 4 : 1;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < p.splitk; ++s) {
            if (vec) {
                const f32x4 t = *(const f32x4 *)(p.partial + s * plane + e0);
                v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
            } else {
                v[0] += p.partial[s * plane + e0];
            }
        }
        const long ib = e0 / mn, rem = e0 - ib * mn;
        const long row = rem / p.n, col = rem - row * p.n;
        unsigned short o[4];
        for (int r = 0; r < cnt; ++r) {
            float x = v[r];
            if (bias)
                x += Tr::to_f32(bias[ib * p.bias_b + row * p.bias_m + (col + r) * p.bias_n]);
            o[r] = Tr::from_f32(apply_act(x, p.act));
        }
        if (vec) {
            u32x2_t pk;
            pk[0] = (unsigned)o[0] | ((unsigned)o[1] << 16);
            pk[1] = (unsigned)o[2] | ((unsigned)o[3] << 16);
            *(u32x2_t *)(C + ib * p.c_bs + c_off(p, row, col)) = pk;
        } else {
            C[ib * p.c_bs + c_off(p, row, col)] = o[0];
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
    // per-lane DMA offsets are 32-bit byte offsets inside one operand matrix
    if ((long)p.m * p.k >= (1l << 31) || (long)p.n * p.k >= (1l << 31))
        return false;
    return true;
}

// Split-K launch: `splits` workgroups per output tile + one reduce pass. For shapes whose 256^2 tiles cannot fill the
// 256 CUs (a 2048-token activation times a 4096-wide weight is 128 tiles) but whose K is long.
template <typename Tr> static int launch256_splitk(infiniRocmRuntime_t rt, GemmArgs p, bool akm, bool bkm, int splits) {
    p.tiles_m = (int)ceil_div(p.m, g256::BM);
    p.tiles_n = (int)ceil_div(p.n, g256::BN);
    const int nk = p.k / g256::BK;
    const int per = (nk + splits - 1) / splits;
    splits = (nk + per - 1) / per; // no empty slice
    void *ws = nullptr;
    const size_t bytes = (size_t)splits * p.batch * p.m * p.n * sizeof(float);
    int st = infini_rocm_workspace(rt, bytes, &ws);
    if (st != INFINI_ROCM_OK)
        return st;
    p.splitk = splits;
    p.partial = (float *)ws;
    const unsigned grid = (unsigned)p.tiles_m * p.tiles_n * p.batch * splits;
#define IROCM_G256S(AK, BK_)                                                                       \
    do {                                                                                           \
        auto kern = g256::gemm256_kernel<Tr, AK, BK_, true>;                                    \
        IROCM_LDS_ATTR(kern, g256::LDS_BYTES, rt);                                                 \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), g256::LDS_BYTES, rt->stream, p);           \
    } while (0)
    if (akm && bkm) IROCM_G256S(true, true);
    else if (akm && !bkm) IROCM_G256S(true, false);
    else if (!akm && bkm) IROCM_G256S(false, true);
    else IROCM_G256S(false, false);
#undef IROCM_G256S
    IROCM_LAUNCH_CHECK("gemm256_splitk");
    const long items = (long)p.batch * p.m * p.n / ((p.n % 4 == 0) ? 4 : 1);
    long g = ceil_div(items, 256);
    if (g > (long)rt->num_cu * 16) g = (long)rt->num_cu * 16;
    hipLaunchKernelGGL(g256::splitk_reduce_kernel<Tr>, dim3((unsigned)g), dim3(256), 0, rt->stream, p);
    IROCM_LAUNCH_CHECK("splitk_reduce");
    return INFINI_ROCM_OK;
}

int launch_gemm256_splitk(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, bool akm, bool bkm, int splits) {
    return dtype == INFINI_DT_BF16 ? launch256_splitk<Bf16Traits>(rt, p, akm, bkm, splits)
                                   : launch256_splitk<F16Traits>(rt, p, akm, bkm, splits);
}

// fp32 output from the split-K planes (the reduced-precision compute types of an fp32 MatMul, gemm.hip): out = act(sum_s
// partial[s] + bias) with an fp32 bias, written as fp32. One thread per 4 consecutive columns when n % 4 == 0.
__global__ __launch_bounds__(256) void splitk_reduce_f32_kernel(GemmArgs p) {
    const long mn = (long)p.m * p.n, total = (long)p.batch * mn;
    const float *bias = (const float *)p.bias;
    float *C = (float *)p.c;
    const bool vec = (p.n % 4 == 0);
    const long items = vec ? total / 4 : total;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const long e0 = vec ? it * 4 : it;
        const int cnt = vec ? 4 : 1;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < p.splitk; ++s) {
            if (vec) {
                const f32x4 t = *(const f32x4 *)(p.partial + s * total + e0);
                v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
            } else {
                v[0] += p.partial[s * total + e0];
            }
        }
        const long ib = e0 / mn, rem = e0 - ib * mn;
        const long row = rem / p.n, col = rem - row * p.n;
        for (int r = 0; r < cnt; ++r) {
            float x = v[r];
            if (bias)
                x += bias[ib * p.bias_b + row * p.bias_m + (col + r) * p.bias_n];
            C[ib * p.c_bs + row * p.n + col + r] = apply_act(x, p.act);
        }
    }
}

// 16-bit operands (p.a / p.b), fp32 accumulation, fp32 OUTPUT (p.c): the split-K form of the 256^2 kernel whose slices write
// raw fp32 sums. One slice, no bias, no activation: the single plane IS the result (the kernel writes p.c directly); otherwise
// the planes go to `planes` (splits * batch * m * n floats, from the caller's workspace carve) and the fp32 reduce follows.
template <typename Tr> static int launch256_f32out(infiniRocmRuntime_t rt, GemmArgs p, bool akm, bool bkm, int splits, float *planes) {
    p.tiles_m = (int)ceil_div(p.m, g256::BM);
    p.tiles_n = (int)ceil_div(p.n, g256::BN);
    const int nk = p.k / g256::BK;
    const int per = (nk + splits - 1) / splits;
    splits = (nk + per - 1) / per;
    const bool direct = splits == 1 && !p.bias && p.act == 0 && p.c_bs == (long)p.m * p.n;
    p.splitk = splits;
    p.partial = direct ? (float *)p.c : planes;
    const unsigned grid = (unsigned)p.tiles_m * p.tiles_n * p.batch * splits;
#define IROCM_G256F(AK, BK_)                                                                       \
    do {                                                                                           \
        auto kern = g256::gemm256_kernel<Tr, AK, BK_, true>;                                       \
        IROCM_LDS_ATTR(kern, g256::LDS_BYTES, rt);                                                 \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), g256::LDS_BYTES, rt->stream, p);           \
    } while (0)
    if (akm && bkm) IROCM_G256F(true, true);
    else if (akm && !bkm) IROCM_G256F(true, false);
    else if (!akm && bkm) IROCM_G256F(false, true);
    else IROCM_G256F(false, false);
#undef IROCM_G256F
    IROCM_LAUNCH_CHECK("gemm256_f32out");
    if (direct)
        return INFINI_ROCM_OK;
    const long items = (long)p.batch * p.m * p.n / ((p.n % 4 == 0) ? 4 : 1);
    long g = ceil_div(items, 256);
    if (g > (long)rt->num_cu * 16) g = (long)rt->num_cu * 16;
    hipLaunchKernelGGL(splitk_reduce_f32_kernel, dim3((unsigned)g), dim3(256), 0, rt->stream, p);
    IROCM_LAUNCH_CHECK("splitk_reduce_f32");
    return INFINI_ROCM_OK;
}

int launch_gemm256_f32out(infiniRocmRuntime_t rt, int dtype16, const GemmArgs &p, bool akm, bool bkm, int splits, float *planes) {
    return dtype16 == INFINI_DT_BF16 ? launch256_f32out<Bf16Traits>(rt, p, akm, bkm, splits, planes)
                                     : launch256_f32out<F16Traits>(rt, p, akm, bkm, splits, planes);
}

template <typename Tr> static int launch256(infiniRocmRuntime_t rt, GemmArgs p, bool akm, bool bkm) {
    p.tiles_m = (int)ceil_div(p.m, g256::BM);
    p.tiles_n = (int)ceil_div(p.n, g256::BN);
    const unsigned grid = (unsigned)p.tiles_m * p.tiles_n * p.batch;
#define IROCM_G256(AK, BK_)                                                                        \
    do {                                                                                           \
        auto kern = g256::gemm256_kernel<Tr, AK, BK_>;                                      \
        IROCM_LDS_ATTR(kern, g256::LDS_BYTES, rt);                                                 \
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
