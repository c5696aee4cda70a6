// 16-bit GEMM, FOUR waves per workgroup x 128 x 128 wave tiles (round 6). Reference contract: src/kernels/cuda/matmul.cc:67-174
// (C = op(A) op(B), fp32 accumulation, 16-bit output); this kernel serves the plain case — no bias, no activation, M and N multiples
// of 256, K a multiple of 128 — everything else stays on gemm256p_kernel.h.
//
// Why a second K loop. The 8-wave kernel (2 waves per SIMD, 128 x 64 wave tiles, LOAD | COMPUTE phases) holds 0.57-0.60 of the nominal
// MFMA peak, and round 6's probes (probe.hip, profiles/r06_mfma_ceiling.json) showed what a wave that shares nothing could do: with ONE
// wave per SIMD a wave owns one issue slot in four and an MFMA (16 cycles) needs one slot in four, so every other instruction rides in
// a free slot IF it stands alone between two MFMAs — a clump of eight of them behind a group of MFMAs drains the matrix pipe (probe:
// 1 450 TF/s clumped, 1 715-1 742 spread, 2 050 MFMA-only on the same box). A 128 x 128 wave tile also reads half the fragment bytes
// per MFMA of a 128 x 64 one. This kernel is that schedule made real:
//   * a ring of FOUR 32 KB stages in LDS, one k-step of 32 each: [A 256 rows x 32 k | B 32 k x 256 columns];
//   * per k-step and wave: 64 MFMAs (8 x 8 tiles of v_mfma_f32_16x16x32), the fragment reads of the NEXT k-step (stage s + 1 into the
//     other register set), 8 LDS-DMA pieces of the k-step THREE ahead (stage s + 3: the stage whose reads ended at the last barrier),
//     s_waitcnt vmcnt(8) — the pieces issued one step ago — and one barrier;
//   * inside a group of eight MFMAs (accumulator row i): A reads behind MFMA 0 / 1, B reads behind MFMA 2 / 3, s_add m0 behind MFMA 4,
//     the piece behind MFMA 6 — inline asm throughout, accumulators pinned to AGPRs ("+a"), fragments to VGPRs;
//   * persistent over tiles (XCD-chunked, 4-row groups), the k-step stream runs THROUGH tile boundaries: the last three steps of a tile
//     request the next tile's first three, the epilogue (permlane16 swap -> 16-byte stores) runs under them.
// LDS images. K-major operand (k contiguous in memory): 16-row blocks of sixteen 64-byte SLOTS; row l15 of a block sits in slot
// s = b0 | b3 << 1 | (b2 b1) << 2 of its bits and physical 16-byte chunk c' of slot s holds logical chunk c' ^ (s >> 2). The bit
// permutation is what makes ds_read_b128 conflict-free: the counters (SQ_LDS_BANK_CONFLICT) showed that a b128 read is served in four
// passes of the lanes with EQUAL (l15 >> 1) & 3 — rows l15 = {2p, 2p + 1, 2p + 8, 2p + 9} of all four lane groups — and with rows in
// plain order those four rows cover only two of the four 64-byte bank quarters (a 2-way conflict on every read: + 4 cycles). M/N-major
// operand: gemm256_common.h's image [32 k][512 B] with the mn_f XOR, read by ds_read_b64_tr_b16 pairs.
#include "gemm256_common.h"

namespace irocm {
namespace g128w {

using g256::mn_f;
using g256::sfor;

constexpr int kStage = 32768, kOper = 16384;
constexpr int kLds = 4 * kStage; // 128 KiB
constexpr int kGroupM = 4;       // tile rows per group: 32 consecutive tile indices (one XCD's CUs) form a 4 x 8 block

struct WArgs {
    GemmArgs g;
    int total_tiles;
    int dbg; // IROCM_W128_DBG (bring-up / diagnosis): 1 = no stores, 2 = every piece reads the tile corner, 8 = clock stamps into C[0..15]
};

// Fragment sets: two register sets x 8 fragments of one operand.
template <bool KMAJOR> struct Frags;
template <> struct Frags<true> {
    s16x8_t v[2][8];
    unsigned base[2]; // lane address in stage 0 / stage 2 (stages 1 / 3 by the immediate offset)
    __device__ __forceinline__ void init(unsigned lds_oper, int half, int l15, int g4) {
        const int slot = (l15 & 1) | (((l15 >> 3) & 1) << 1) | (((l15 >> 1) & 3) << 2); // row l15 of a 16-row block -> its 64-byte slot
        base[0] = lds_oper + (unsigned)(half * 8192 + slot * 64 + ((g4 ^ (slot >> 2)) & 3) * 16);
        base[1] = base[0] + 2u * kStage;
    }
    template <int SET, int F, int STAGE> __device__ __forceinline__ void read0() {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[SET][F]) : "v"(base[STAGE >> 1]), "i"(F * 1024 + (STAGE & 1) * kStage));
    }
    template <int SET, int F, int STAGE> __device__ __forceinline__ void read1() {}
    template <int SET, int F> __device__ __forceinline__ s16x8_t get() const { return v[SET][F]; }
};
template <> struct Frags<false> {
    s16x4_t lo[2][8], hi[2][8];
    unsigned addr[8]; // lane address of fragment f in stage 0 (stage 1 by the immediate offset, stages 2 / 3 by one VALU add at the read:
                      // sixteen more address registers per operand spilled the both-operands-M/N-major build)
    __device__ __forceinline__ void init(unsigned lds_oper, int half, int l15, int g4) {
        const int mnf = ((l15 >> 2) & 3) | ((g4 & 1) << 2);
        const unsigned mn_lane = (unsigned)((g4 * 8 + (l15 >> 2)) * 512 + (l15 & 1) * 8);
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const int c16 = ((l15 >> 1) & 1) | (((half * 8 + f) ^ mnf) << 1);
            addr[f] = lds_oper + mn_lane + (unsigned)c16 * 16u;
        }
    }
    template <int SET, int F, int STAGE> __device__ __forceinline__ void read0() {
        const unsigned a = STAGE >= 2 ? addr[F] + 2u * kStage : addr[F];
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[SET][F]) : "v"(a), "i"((STAGE & 1) * kStage));
    }
    template <int SET, int F, int STAGE> __device__ __forceinline__ void read1() {
        const unsigned a = STAGE >= 2 ? addr[F] + 2u * kStage : addr[F];
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[SET][F]) : "v"(a), "i"((STAGE & 1) * kStage + 2048));
    }
    template <int SET, int F> __device__ __forceinline__ s16x8_t get() const {
        return s16x8_t{lo[SET][F][0], lo[SET][F][1], lo[SET][F][2], lo[SET][F][3], hi[SET][F][0], hi[SET][F][1], hi[SET][F][2], hi[SET][F][3]};
    }
};

// per-lane byte offsets of a wave's four pieces of one operand's stage, relative to the tile's (row0 / col0, k0) corner
template <bool KMAJOR> __device__ __forceinline__ void piece_offs(unsigned (&off)[4], long ld, int w, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = w * 4 + i;
        if constexpr (KMAJOR) {
            const int slot = lane >> 2; // the lane's 64-byte slot of the piece's 1 KB; it holds row (bit permutation below) of the block
            const int r = piece * 16 + ((slot & 1) | (((slot >> 2) & 3) << 1) | (((slot >> 1) & 1) << 3));
            const int c_log = (lane & 3) ^ ((slot >> 2) & 3);
            off[i] = (unsigned)(((long)r * ld + c_log * 8) * 2);
        } else {
            const int kr = piece * 2 + (lane >> 5);
            const int c_log = (lane & 31) ^ (mn_f(kr) << 1);
            off[i] = (unsigned)(((long)kr * ld + c_log * 8) * 2);
        }
    }
}

// Which pieces a wave requests in the k-step at position S of a block of four (the pieces of k-step g + 3, stage (S + 3) & 3).
// An M/N-major operand takes four every step. A K-MAJOR operand's 64-byte row segment is HALF a 128-byte cache line whose other half
// is the next k-step's: requested a step apart, every line would travel L2 -> L1 twice (measured: each K-major operand cost 5-10 %).
// So a K-major operand requests on ODD steps only, the four pieces of k-step g + 3 each followed by its sibling of k-step g + 4 (same
// lane offset + 64 bytes, into stage S & 3 — the stage the MFMAs of this step took their fragments from a step ago, free since the
// last barrier): the sibling finds its line in flight or in L1.
struct PieceSel {
    int oper, idx, stage, imm;
};
template <bool AKM, bool BKM> constexpr int step_pieces(int S) { return (AKM ? ((S & 1) ? 8 : 0) : 4) + (BKM ? ((S & 1) ? 8 : 0) : 4); }
template <bool AKM, bool BKM> constexpr PieceSel piece_sel(int S, int q) {
    const int na = AKM ? ((S & 1) ? 8 : 0) : 4;
    const int oper = q < na ? 0 : 1, r = q < na ? q : q - na;
    const bool km = oper == 0 ? AKM : BKM;
    if (!km)
        return {oper, r, (S + 3) & 3, 0};
    return {oper, r >> 1, (r & 1) ? (S & 3) : ((S + 3) & 3), (r & 1) * 64};
}
// entry of the list that goes behind MFMA 4 / 5 (slot 0) or 6 / 7 (slot 1) of MFMA group I; -1: none
constexpr int slot_piece(int n, int I, int slot) {
    if (n == 0)
        return -1;
    if (n <= 8) {
        const int stride = 8 / n;
        return (slot == 0 && I % stride == 0) ? I / stride : -1;
    }
    return slot == 0 ? I : (8 + I < n ? 8 + I : -1);
}

template <typename Tr, bool FIRST> __device__ __forceinline__ void mfma_a(f32x4 &acc, s16x8_t x, s16x8_t y) {
    if constexpr (Tr::kDType == INFINI_DT_BF16) {
        if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(x), "v"(y));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(x), "v"(y));
    } else {
        if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(acc) : "v"(x), "v"(y));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(x), "v"(y));
    }
}

template <typename Tr, bool AKM, bool BKM>
__global__ __launch_bounds__(256, 1) void gemm128w_kernel(WArgs pw) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const GemmArgs &p = pw.g;
    const int t = threadIdx.x, lane = t & 63, l15 = lane & 15, g4 = lane >> 4;
    int w;
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(w) : "v"(t >> 6));
    const int wr = w >> 1, wc = w & 1;
    const unsigned lds0 = (unsigned)(unsigned long)IROCM_LDS_PTR(smem);
    unsigned lds0s;
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(lds0s) : "v"(lds0));

    // ---- operand geometry (bytes) -----------------------------------------------------------------
    const long lda = AKM ? p.a_rs : p.a_cs, ldb = BKM ? p.b_cs : p.b_rs; // elements between rows of the image's major index
    unsigned a_kstep = (unsigned)(AKM ? 64 : 64 * lda), b_kstep = (unsigned)(BKM ? 64 : 64 * ldb); // 32 k
    const long a_tile = AKM ? 512 * lda : 512, b_tile = BKM ? 512 * ldb : 512; // 256 rows / columns
    // The k advance lives in the LANE offsets (eight VALU adds per step), the 64-bit corner in an SGPR pair that changes once per tile:
    // a VMEM instruction in inline asm must not read an SGPR within five wait states of the instruction that wrote it, and a pointer the
    // compiler advances by SALU every step ends up written wherever its scheduler likes.
    unsigned a_off[4], b_off[4];
    piece_offs<AKM>(a_off, lda, w, lane);
    piece_offs<BKM>(b_off, ldb, w, lane);
    if (pw.dbg & 2) {
        a_kstep = b_kstep = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) a_off[i] = b_off[i] = (unsigned)lane * 16u;
    }
    const int ksteps = p.k >> 5, nkb = ksteps >> 2;
    const unsigned a_span = a_kstep * (unsigned)ksteps, b_span = b_kstep * (unsigned)ksteps; // a tile's whole k advance

    // ---- tiles: workgroup ids consecutive per XCD, tile index -> (batch, tile row, tile column) in groups of kGroupM rows -------
    const unsigned grid = gridDim.x, wg = xcd_remap(blockIdx.x, grid);
    const unsigned per_batch = (unsigned)(p.tiles_m * p.tiles_n);
    auto decode = [&](unsigned tile, const char *&ab, const char *&bb, long &c_elem) {
        const unsigned ib = tile / per_batch, r = tile - ib * per_batch;
        const unsigned gspan = (unsigned)(kGroupM * p.tiles_n), grp = r / gspan, i = r - grp * gspan;
        const unsigned first_m = grp * kGroupM;
        const unsigned gsz = (unsigned)p.tiles_m - first_m < (unsigned)kGroupM ? (unsigned)p.tiles_m - first_m : (unsigned)kGroupM;
        const unsigned tn = i / gsz, tm = first_m + (i - tn * gsz);
        ab = (const char *)p.a + ((long)ib * p.a_bs) * 2 + (long)tm * a_tile;
        bb = (const char *)p.b + ((long)ib * p.b_bs) * 2 + (long)tn * b_tile;
        c_elem = (long)ib * p.c_bs + (long)tm * 256 * p.n + (long)tn * 256;
    };
    auto uniform = [](const char *q) {
        const unsigned long v = (unsigned long)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const char *)(((unsigned long)hi << 32) | lo);
    };

    unsigned tile = wg;
    const char *pa, *pb; // tile corner the pieces of the NEXT issuing step read from (a_off / b_off carry the k advance)
    long c_elem;
    decode(tile, pa, pb, c_elem);
    pa = uniform(pa);
    pb = uniform(pb);

    // ---- LDS-DMA ------------------------------------------------------------------------------------
    unsigned m0w = lds0s + (unsigned)w * 4096u; // this wave's first piece slot of operand A in stage 0
    auto load_piece = [&](auto operc, auto idxc, auto immc) __attribute__((always_inline)) {
        constexpr int OP = decltype(operc)::value, IDX = decltype(idxc)::value, IMM = decltype(immc)::value;
        const unsigned off = OP == 0 ? a_off[IDX] : b_off[IDX]; // (copies: clang does not capture a variable a generic lambda names
        const char *base = OP == 0 ? pa : pb;                   //  only in an asm operand)
        asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(off), "s"(base), "i"(IMM) : "memory");
    };
    // (the instruction offset of an LDS-DMA load moves BOTH addresses, memory and LDS: IMM is taken back out of M0)
    auto set_m0 = [&m0w](auto stc, auto operc, auto idxc, auto immc) __attribute__((always_inline)) {
        constexpr int ST = decltype(stc)::value, OP = decltype(operc)::value, IDX = decltype(idxc)::value, IMM = decltype(immc)::value;
        const unsigned m0base = m0w;
        // (s_add writes SCC: undeclared, the compiler parks an add-with-carry or a compare across this statement)
        asm volatile("s_add_u32 m0, %0, %1" ::"s"(m0base), "i"(ST * kStage + OP * kOper + IDX * 1024 - IMM) : "memory", "scc");
    };
    // A VMEM instruction must not read an SGPR within five wait states of the SALU / VALU instruction that wrote it, and the compiler
    // keeps that rule only for instructions it knows: the tile corners are therefore pinned into their registers where they are
    // computed (an empty asm the pointer passes through), at least one MFMA group away from the inline-asm pieces that read them.
    auto pin = [&]() __attribute__((always_inline)) { asm volatile("" : "+s"(pa), "+s"(pb)); };
    // prologue: one operand's four pieces of k-step ST (M/N-major) or of the k-step pair ST, ST + 1 (K-major), back to back
    auto request = [&](auto stc, auto operc) __attribute__((always_inline)) {
        constexpr int ST = decltype(stc)::value, OP = decltype(operc)::value;
        constexpr bool KM = OP == 0 ? AKM : BKM;
        sfor<4>([&](auto idxc) {
            set_m0(stc, operc, idxc, std::integral_constant<int, 0>{});
            asm volatile("s_nop 0" ::: "memory");
            load_piece(operc, idxc, std::integral_constant<int, 0>{});
            if constexpr (KM) {
                set_m0(std::integral_constant<int, ST + 1>{}, operc, idxc, std::integral_constant<int, 64>{});
                asm volatile("s_nop 0" ::: "memory");
                load_piece(operc, idxc, std::integral_constant<int, 64>{});
            }
        });
    };
    auto advance = [&](auto operc, unsigned by) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (decltype(operc)::value == 0) a_off[i] += by;
            else b_off[i] += by;
        }
    };
    constexpr std::integral_constant<int, 0> kA{};
    constexpr std::integral_constant<int, 1> kB{};

    // ---- fragments -----------------------------------------------------------------------------------
    Frags<AKM> fa;
    Frags<BKM> fb;
    fa.init(lds0, wr, l15, g4);
    fb.init(lds0 + kOper, wc, l15, g4);
    f32x4 acc[8][8];

    // ---- prologue: k-steps 0 .. 2 of the first tile requested (a K-major operand: pairs 0 and 1 = k-steps 0 .. 3), step 0's fragments
    // read. Issue order = order of need, so that the counted waits below leave exactly the later groups in flight.
    constexpr int nKM = (AKM ? 1 : 0) + (BKM ? 1 : 0), nMN = 2 - nKM;
    pin();
    asm volatile("s_nop 4" ::: "memory");
    if constexpr (AKM) request(std::integral_constant<int, 0>{}, kA); // group 0: everything k-step 0 needs (K-major: the pair 0, 1)
    if constexpr (BKM) request(std::integral_constant<int, 0>{}, kB);
    if constexpr (!AKM) request(std::integral_constant<int, 0>{}, kA);
    if constexpr (!BKM) request(std::integral_constant<int, 0>{}, kB);
    if constexpr (AKM) advance(kA, 128);
    else advance(kA, a_kstep);
    if constexpr (BKM) advance(kB, 128);
    else advance(kB, b_kstep);
    if constexpr (!AKM) request(std::integral_constant<int, 1>{}, kA); // group 1: the M/N-major operands' k-step 1
    if constexpr (!BKM) request(std::integral_constant<int, 1>{}, kB);
    if constexpr (!AKM) advance(kA, a_kstep);
    if constexpr (!BKM) advance(kB, b_kstep);
    if constexpr (AKM) request(std::integral_constant<int, 2>{}, kA); // group 2: k-step 2 (K-major: the pair 2, 3)
    if constexpr (BKM) request(std::integral_constant<int, 2>{}, kB);
    if constexpr (!AKM) request(std::integral_constant<int, 2>{}, kA);
    if constexpr (!BKM) request(std::integral_constant<int, 2>{}, kB);
    if constexpr (AKM) advance(kA, 128);
    else advance(kA, a_kstep);
    if constexpr (BKM) advance(kB, 128);
    else advance(kB, b_kstep);
    constexpr int kG1 = 4 * nMN, kG2 = 8 * nKM + 4 * nMN;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kG1 + kG2) : "memory");
    __builtin_amdgcn_s_barrier();
    sfor<8>([&](auto fc) {
        constexpr int F = decltype(fc)::value;
        fa.template read0<0, F, 0>();
        fa.template read1<0, F, 0>();
        fb.template read0<0, F, 0>();
        fb.template read1<0, F, 0>();
    });
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kG2) : "memory");
    __builtin_amdgcn_s_barrier();

    // one k-step: S = position in the block of four (register set S & 1; reads stage S + 1; its pieces go to stage S + 3)
    const char *na = pa, *nb = pb; // the next tile's corner (this tile's when there is none: requested, never read)
    auto step = [&](auto sc, auto firstc, bool last_block) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value;
        constexpr bool FIRST = decltype(firstc)::value;
        constexpr int CUR = S & 1, NXT = CUR ^ 1, RS = (S + 1) & 3, NP = step_pieces<AKM, BKM>(S);
        sfor<8>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            sfor<8>([&](auto jc) {
                constexpr int J = decltype(jc)::value;
                mfma_a<Tr, FIRST>(acc[I][J], fb.template get<CUR, J>(), fa.template get<CUR, I>());
                if constexpr (J == 0) fa.template read0<NXT, I, RS>();
                if constexpr (J == 1) fa.template read1<NXT, I, RS>();
                if constexpr (J == 2) fb.template read0<NXT, I, RS>();
                if constexpr (J == 3) fb.template read1<NXT, I, RS>();
                if constexpr (J >= 4) {
                    constexpr int Q = slot_piece(NP, I, (J - 4) >> 1);
                    if constexpr (Q >= 0) {
                        constexpr PieceSel ps = piece_sel<AKM, BKM>(S, Q);
                        if constexpr ((J & 1) == 0)
                            set_m0(std::integral_constant<int, ps.stage>{}, std::integral_constant<int, ps.oper>{}, std::integral_constant<int, ps.idx>{},
                                   std::integral_constant<int, ps.imm>{});
                        else
                            load_piece(std::integral_constant<int, ps.oper>{}, std::integral_constant<int, ps.idx>{}, std::integral_constant<int, ps.imm>{});
                    }
                }
            });
        });
        // the pieces of the steps after this one: the next k-step (pair) of this tile, or — behind the tile's fourth-last step, where both
        // kinds of operand have requested everything of this tile — the next tile's first
        if (S == 0 && last_block) {
            pa = na;
            pb = nb;
            pin();
            advance(kA, AKM ? 0u - a_span : a_kstep - a_span);
            advance(kB, BKM ? 0u - b_span : b_kstep - b_span);
        } else {
            if constexpr (!AKM) advance(kA, a_kstep);
            else if constexpr (S & 1) advance(kA, 128);
            if constexpr (!BKM) advance(kB, b_kstep);
            else if constexpr (S & 1) advance(kB, 128);
        }
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NP) : "memory");
        __builtin_amdgcn_s_barrier();
    };

    const unsigned long long tc0 = __builtin_amdgcn_s_memtime(), tr0 = __builtin_amdgcn_s_memrealtime();
    const unsigned total = (unsigned)pw.total_tiles;
    const long ldc2 = (long)p.n * 2;
    for (; tile < total; tile += grid) {
        {
            const unsigned nt = tile + grid < total ? tile + grid : tile;
            long dummy;
            decode(nt, na, nb, dummy);
            na = uniform(na);
            nb = uniform(nb);
        }
        {
            const bool last = nkb == 1;
            step(std::integral_constant<int, 0>{}, std::true_type{}, last);
            step(std::integral_constant<int, 1>{}, std::false_type{}, last);
            step(std::integral_constant<int, 2>{}, std::false_type{}, last);
            step(std::integral_constant<int, 3>{}, std::false_type{}, last);
        }
        for (int kb = 1; kb < nkb; ++kb) {
            const bool last = kb == nkb - 1;
            step(std::integral_constant<int, 0>{}, std::false_type{}, last);
            step(std::integral_constant<int, 1>{}, std::false_type{}, last);
            step(std::integral_constant<int, 2>{}, std::false_type{}, last);
            step(std::integral_constant<int, 3>{}, std::false_type{}, last);
        }
        // ---- epilogue: lane l holds, of tile (i, j), row l15 and the four columns g4 * 4 ...; a permlane16 swap between the packed
        // halves of tiles j and j + 1 leaves every lane eight consecutive columns: one 16-byte store per tile pair ------------------
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); // the last MFMAs' results (inline asm: no hazard bookkeeping by the compiler)
        // (the lane's store offset is rebuilt from mbcnt here: carried across the K loop it is the register the M/N x M/N build spills)
        const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const unsigned c_lane = (unsigned)(((long)(ln & 15) * p.n + ((ln >> 4) & 1) * 16 + (ln >> 5) * 8) * 2);
        char *cw = (char *)p.c + (c_elem + (long)wr * 128 * p.n + wc * 128) * 2;
        cw = (char *)uniform(cw);
        if (!(pw.dbg & 1))
        sfor<8>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            char *ci = cw + (long)I * 16 * ldc2;
            sfor<4>([&](auto jpc) {
                constexpr int JP = decltype(jpc)::value;
                const f32x4 x = acc[I][2 * JP], y = acc[I][2 * JP + 1];
                unsigned u0 = Tr::pack2(x[0], x[1]), u1 = Tr::pack2(x[2], x[3]);
                unsigned v0 = Tr::pack2(y[0], y[1]), v1 = Tr::pack2(y[2], y[3]);
                asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(u0), "+v"(v0));
                asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(u1), "+v"(v1));
                // (asm: a store the compiler knows about makes it guard the tile loop with s_waitcnt vmcnt(0), i.e. wait out the next
                // tile's pieces; a store has read its registers once it has issued)
                const u32x4_t d = u32x4_t{u0, u1, v0, v1};
                const unsigned cl = c_lane; // (copies: clang does not capture a variable a generic lambda names only in an asm operand)
                char *cb = ci;
                if constexpr (JP == 0) asm volatile("s_nop 4" : "+s"(cb)); // (the row block's base was just computed by SALU)
                // (s_nop: a store of more than 64 bits must be two wait states ahead of a VALU write of its data registers — a rule
                // the compiler keeps only for stores it knows)
                asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" ::"v"(cl), "v"(d), "s"(cb), "i"(JP * 64) : "memory");
            });
        });
        // this workgroup's next tile
        {
            const char *ta, *tb;
            const unsigned nt = tile + grid < total ? tile + grid : tile;
            decode(nt, ta, tb, c_elem);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // pieces requested past the last tile
    if ((pw.dbg & 8) && blockIdx.x == 0 && w == 0 && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) { // tools/gemm_wave128.py --clock: the K loops' length on the core clock and on the 100 MHz
        unsigned long long *d = (unsigned long long *)p.c; // reference, over the first 16 bytes of C
        d[0] = __builtin_amdgcn_s_memtime() - tc0;
        d[1] = __builtin_amdgcn_s_memrealtime() - tr0;
    }
}

bool supported(const GemmArgs &p) {
    if (p.m <= 0 || p.n <= 0 || p.k < 128 || p.m % 256 || p.n % 256 || p.k % 128)
        return false;
    if (p.bias || p.act != 0 || p.hs_d != 0 || p.splitk > 1)
        return false;
    if ((((uintptr_t)p.a) | ((uintptr_t)p.b) | ((uintptr_t)p.c)) & 15)
        return false;
    if ((p.a_bs % 8) || (p.b_bs % 8) || (p.c_bs % 8))
        return false;
    // per-lane piece offsets are 32-bit: 256 rows of the major index must stay below 4 GiB
    const long lda = p.a_cs == 1 ? p.a_rs : p.a_cs, ldb = p.b_rs == 1 ? p.b_cs : p.b_rs;
    if (lda * 512 >= (1ll << 32) || ldb * 512 >= (1ll << 32) || (long)p.n * 32 >= (1ll << 32))
        return false;
    const long tiles = (long)(p.m / 256) * (p.n / 256) * p.batch;
    return tiles < (1ll << 31);
}

template <typename Tr> static int launch_t(infiniRocmRuntime_t rt, GemmArgs p, bool akm, bool bkm) {
    WArgs w;
    p.tiles_m = p.m / 256;
    p.tiles_n = p.n / 256;
    w.g = p;
    w.total_tiles = p.tiles_m * p.tiles_n * p.batch;
    const char *dbg = getenv("IROCM_W128_DBG");
    w.dbg = dbg ? atoi(dbg) : 0;
    const unsigned grid = (unsigned)(w.total_tiles < rt->num_cu ? w.total_tiles : rt->num_cu);
#define IROCM_G128W(AK, BK_)                                                                                         \
    do {                                                                                                             \
        auto kern = gemm128w_kernel<Tr, AK, BK_>;                                                                    \
        IROCM_LDS_ATTR(kern, kLds, rt);                                                                              \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kLds, rt->stream, w);                                        \
    } while (0)
    if (akm && bkm) IROCM_G128W(true, true);
    else if (akm && !bkm) IROCM_G128W(true, false);
    else if (!akm && bkm) IROCM_G128W(false, true);
    else IROCM_G128W(false, false);
#undef IROCM_G128W
    IROCM_LAUNCH_CHECK("gemm128w");
    return INFINI_ROCM_OK;
}

int launch_gemm128w(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, bool akm, bool bkm) {
    return dtype == INFINI_DT_BF16 ? launch_t<Bf16Traits>(rt, p, akm, bkm) : launch_t<F16Traits>(rt, p, akm, bkm);
}

} // namespace g128w
} // namespace irocm
