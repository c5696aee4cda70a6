// gemm256p: the PERSISTENT, multi-tile edition of the staggered 256-row GEMM of gemm256.hip (SCHED 0).
//
// Why: a 256 x 256 x 64 workgroup pays ~14.5 us of fixed cost per output tile (cold first K-tiles ~5 us, the C store
// tail ~7 us, launch ~2 us: tools/gemm_fixed_cost.py) around a main loop of 1.44 us per K-tile. One tile per workgroup
// leaves that uncovered whenever K is short (BERT: 12 K-tiles) or a problem has several tiles per CU. Here ONE
// workgroup per CU walks its tiles (tile ids blockIdx.x, + gridDim.x, ...) through ONE flat K-tile pipeline:
//   * the LDS-DMA cursors run across tile boundaries (A one K-tile ahead, B two), so the next tile's first two K-tiles
//     are in LDS before the current tile's last MFMA: no cold prologue after the first tile;
//   * the epilogue of tile T sits between C2 of T's last K-tile and L1 of the next tile's first one; C stores are
//     asynchronous and drain under the next tile's main loop. The two wave rows run their epilogues SIDE BY SIDE: row 0,
//     which is one barrier interval ahead (the LOAD | COMPUTE stagger), first waits out row 1's last compute interval and
//     row 1 falls back one interval after its own epilogue (two extra barriers per tile, see the tile loop).
//     (A variant that issued the next A DMA before the C stores and counted the stores in the following vmcnt wait was
//     built and removed: loads and stores retire through one counter but not in one order, so a counted wait behind
//     stores proves nothing about older loads; every wait here is a plain vmcnt(N <= loads issued after the one needed),
//     which is safe because loads retire in order among themselves.)
// What the TRACE build measured (tools/gemm_timeline.py, 16384 x 3072 x 768, 3 tiles per workgroup): steady K-tile
// ~3000 cycles (2048 of them MFMA issue); a wave row's epilogue ~9400 cycles. In the first version the epilogue had no
// barrier in it, and the workgroup-wide barriers of the K-tiles around it serialised the rows: row 1's last (short)
// compute interval lasted as long as row 0's epilogue and row 0's first interval of the next tile as long as row 1's,
// 18.7 k cycles per tile boundary — misread at first as a per-CU store limit of 7 bytes / clk (and "confirmed" by
// experiments that could not move it: 8 rows x 128 B instead of 16 x 64 B per store, non-temporal stores, workgroups
// started in phases; all removed). Side by side both rows finish in ~10.5 k cycles: DESIGN.md, "Follow-up".
// Tile width is a template parameter: NT = 4 / 3 / 2 MFMA column tiles per wave = 256 / 192 / 128 columns (8 waves as
// 2 (M) x 4 (N); wave tile 128 x 16 NT), so N = 768 (BERT) tiles as 4 x 192 without a ragged last tile and small
// grids can trade tile size for CU coverage. Everything else — LDS images, XOR swizzles, transpose reads, the
// LOAD | COMPUTE stagger, counted vmcnt, inline-asm ds_reads — is gemm256.hip's; see the comments there.
// N-major B with NT < 4 keeps the 512-byte-row LDS image and fetches 256 columns (the unused ones clamp to valid
// memory): B is the small, L2-resident operand.
#pragma once
#include "gemm256_common.h"

// K-loop schedule variant of the plain-GEMM instantiations (round 6 A/B, tools/gemm_kv_ab.sh builds one library per value; the conv /
// tap modes always run variant 0). Where the LDS-DMA pieces of a LOAD phase are issued relative to its fragment reads:
//   0  reads, DMA, lgkm wait, barrier (rounds 2-5)        1  DMA, reads, wait, barrier
//   2  reads, wait, DMA, barrier                           3  reads and DMA pieces interleaved (8 reads : 2 pieces)
//   4  = 0 with the accumulators in AGPRs                  6  NO DMA in the K loop (wrong sums: timing-only ablation)
//   9  every piece issued from a COMPUTE phase, one behind each group of 8 MFMAs (see burst9 in the kernel)
//  10  = 0 with EVERY piece issued by wave row 0 (8 per LOAD phase), none by row 1
//   7  = 0 with a static s_setprio 1 for wave row 1 instead of per-phase flips      8  = 0 without any s_setprio
#ifndef IROCM_KV
#define IROCM_KV 0
#endif

namespace irocm {
namespace g256p {
using namespace g256;

template <int N> __device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// K-major operand of ROWS = 8 * 8 * CNT rows: 8 * CNT pieces of 8 rows; wave w issues pieces w*CNT .. w*CNT+CNT-1.
template <int CNT>
__device__ __forceinline__ void offs_k_n(unsigned (&off)[4], long ld, int row0, int rows, int w, int lane) {
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const int piece = w * CNT + i;
        const int r = piece * 8 + (lane >> 3);
        const int c_log = (lane & 7) ^ ((r >> 1) & 7);
        int gr = row0 + r;
        gr = gr < rows ? gr : rows - 1;
        off[i] = (unsigned)(((long)gr * ld + c_log * 8) * 2);
    }
}
// offs_k with the eight 16-row blocks of each 128-row half rotated: LDS row block b holds the tile's row block (b + rot) % 8
// (split-K of the tap mode: see exchange() in the kernel)
__device__ __forceinline__ void offs_k_rot(unsigned (&off)[4], long ld, int row0, int rows, int w, int lane, int rot) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = w * 4 + i;
        const int r = piece * 8 + (lane >> 3);
        const int c_log = (lane & 7) ^ ((r >> 1) & 7);
        const int rr = (r & ~0x70) | ((((r >> 4) + rot) & 7) << 4);
        int gr = row0 + rr;
        gr = gr < rows ? gr : rows - 1;
        off[i] = (unsigned)(((long)gr * ld + c_log * 8) * 2);
    }
}
template <int CNT>
__device__ __forceinline__ void stage_n(const char *ubase, const unsigned (&off)[4], char *lds_oper, int w) {
#pragma unroll
    for (int i = 0; i < CNT; ++i)
        __builtin_amdgcn_global_load_lds(IROCM_GLB_PTR(ubase + (unsigned long)off[i]),
                                         IROCM_LDS_PTR(lds_oper + (w * CNT + i) * 1024), 16, 0, 0);
}

// pieces I0 .. I1 - 1 of a wave's CNT (interleaved issue: variant 3)
template <int CNT, int I0, int I1>
__device__ __forceinline__ void stage_part(const char *ubase, const unsigned (&off)[4], char *lds_oper, int w) {
#pragma unroll
    for (int i = I0; i < I1 && i < CNT; ++i)
        __builtin_amdgcn_global_load_lds(IROCM_GLB_PTR(ubase + (unsigned long)off[i]),
                                         IROCM_LDS_PTR(lds_oper + (w * CNT + i) * 1024), 16, 0, 0);
}

struct PArgs {
    GemmArgs g;
    int total_tiles; // tiles_m * tiles_n * batch
    int tab_n;       // steps of a workgroup served by the LDS tile table (kTileTab, or 0: tile counts beyond 16 bits)
    unsigned per_batch_m, per_group_m; // floor(2^32 / (tiles_m * tiles_n)), floor(2^32 / (8 * tiles_n)): decode() divides by multiply-high
    unsigned long long *trace; // TRACE instantiation only: [gridDim.x][8 waves][kTraceSlots] s_memtime stamps
    int trace_fine;            // TRACE: 1 = six stamps per K-tile (L1 | C1 | after C1's MFMAs | L2 | C2 | after C2's MFMAs) instead of two
    // tap mode, split-K (CONV = 3 only; 1 = off): `split` workgroups share ONE output tile, each sums a contiguous range of the
    // tile's K-tiles, then the slices trade accumulator row blocks through `slab` (fp32, written through) and every slice
    // finishes 8 / split of each wave's eight 16-row blocks: see exchange() in the kernel.
    int split;
    char *slab;       // [tile][source slice][wave][row block][column tile][lane] x 16 bytes
    unsigned slab_bytes;
    unsigned *flags;  // [tile][source slice][destination slice][wave]: zero between launches (the consumer resets its flag)
    unsigned *err;    // set (system scope: pinned host memory) by a wave whose bounded wait on a flag ran out — infiniRocmRuntime::sync_err_host
};

constexpr int kTraceSlots = 128;
constexpr int kTraceBytes = 8 * kTraceSlots * 8;
// Decoded tiles (batch, m0, n0) of this workgroup's first kTileTab steps, 16 bytes each, in LDS behind the K-tile buffers (and
// the TRACE strip). Round 4: the scalar decode() — XCD remap, three multiply-high divisions, the group-size ladder: ~120 SALU
// instructions with ~20 branches, ~1.0-1.3 k cycles — ran three times per tile (B cursor, A cursor, tile loop), the first two
// inside a LOAD phase the partner wave row waits for at the barrier: the timeline (tools/gemm_timeline.py) showed +1.3 k cycles
// on exactly the two K-tiles where the cursors cross into the next tile, ~4 k cycles per tile boundary with the third = 8 % of a
// 12-K-tile tile (BERT). Now every thread decodes ONE tile in the prologue (under the first DMA round trip), a cursor that crosses
// fetches its entry with one asm ds_read issued in front of the phase's lgkmcnt wait, and the per-lane offsets are rebuilt
// AFTER the MFMA burst of the following COMPUTE phase, where this wave row has slack.
// An entry is two dwords {batch << 16 | m0 / 256, batch << 16 | n0 / tile width}: the A cursor reads the first, the B cursor the second
// (launch_p / launch_p_conv switch the table off — PArgs::tab_n = 0 — for problems whose batch or tile counts do not fit 16
// bits): ONE VGPR per cursor across a barrier interval; the 256-column bf16 builds sit at 256 VGPRs and a three-dword entry per
// cursor put 24-44 bytes per lane into scratch, a two-dword one 8.
constexpr int kTileTab = 256;
constexpr int kTabBytes = kTileTab * 8;
// Behind the table: two 512-byte slots for the row bias of the tile being accumulated (tile step & 1). The epilogue used to fetch
// its bias with global loads whose wait is a vmcnt(0) — i.e. it also waited out the NEXT tile's first (cold) operand DMAs, already
// in flight: ~1.4 k of the ~2.1 k cycles of the first store step in the timeline. Now wave 0 requests the 64 NT bias values by ONE
// LDS-DMA at the top of the tile (older than every load the K loop's counted waits leave outstanding, so the first of those
// waits covers it) and the epilogue reads them with ds_read_b64.
// (Slots are 1 KiB apart: all 64 lanes of the DMA write 16 bytes each — the upper lanes re-read the tile's last 16 bytes into the
// unused half — so the request needs no lane mask: a divergent branch at the top of the tile loop cost the 256-column builds two
// VGPRs they do not have, 8 bytes of scratch per lane.)
constexpr int kBiasSlotBytes = 1024;
constexpr int kExtraLds = kTabBytes + 2 * kBiasSlotBytes;

// TRACE: every wave stamps s_memtime at phase boundaries into a private LDS strip behind the K-tile buffers (no VMEM
// traffic, so the vmcnt bookkeeping is untouched) and dumps the strip at the end — tools/gemm_timeline.py.
// CONV (round 3): the pointwise-convolution mode described at GemmArgs::cv_hw — the B tile is gathered from an NCHW activation
// by pixel slots (offs_mn_conv), the epilogue adds a per-FILTER bias and an optional residual and stores NCHW (16-byte runs inside
// a plane, the ragged last run of a plane element-wise). A K-major (the FCRS weights of a 1 x 1 layer ARE [F][C]), B N-major.
// CONV = 1: without, CONV = 2: with the residual (its own instantiation: the residual copy of the epilogue needs 16 NT more
// registers, which the 256-column build does not have — it exists for NT <= 3 only).
// CONV = 3 (round 5): TAP mode — a 3 x 3 convolution as the same GEMM with K = 9 C (GemmArgs::cv_taps): the K-tile sequence of a tile
// is (channel block, tap) with the tap inner; the A cursor walks the re-packed weights [tap][F][C], the B cursor adds a per-tap
// byte offset to the pointwise tile's addresses (a tap only MOVES a 16-byte run of pixel slots: unit stride on X itself, stride 2
// on the de-interleaved phase planes), and what the move drags in from the neighbouring row / image / plane — the zero padding —
// is removed by ONE per-lane AND mask per B fragment (a lane's B fragment of v_mfma_f32_16x16x32 is 8 channels of ONE slot):
// 4 NT v_bfe + 8 NT v_and per K-tile beside 16 NT MFMAs. Why here and not in conv_s1.hip's patch kernels: those stream the weight
// slab through 128 x 128 tiles (64 FLOP per L2 byte, 23 % MFMA busy); this loop runs 256 x 256 tiles
// (128 FLOP per byte) through the pipeline that holds 0.58 of the MFMA peak on a plain GEMM.
template <typename Tr, bool A_KMAJOR, bool B_KMAJOR, int NT, bool TRACE = false, int CONV = 0>
__global__ __launch_bounds__(512, 2) void gemm256p_kernel(PArgs pa) {
    const GemmArgs &p = pa.g;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int tslot = 0;
    auto stamp = [&]() {
        if constexpr (TRACE) {
            const unsigned long long tm = __builtin_amdgcn_s_memtime();
            if (tslot < kTraceSlots && (threadIdx.x & 63) == 0)
                *(unsigned long long *)(smem + LDS_BYTES + (threadIdx.x >> 6) * (kTraceSlots * 8) + tslot * 8) = tm;
            ++tslot;
        }
    };
    constexpr int BN_ = 64 * NT;          // tile width
    constexpr int NB = B_KMAJOR ? NT : 4; // B DMA pieces per wave and K-tile
    constexpr int NJ1 = NT - 2;           // column tiles of B sub-tile 1 (sub-tile 0 always has two)
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int l15 = lane & 15, g4 = lane >> 4;

    // split-K (tap mode): workgroup b = 8 idx + xcd works on slice idx % S of tile (idx / S) * 8 + xcd — the S slices of a tile run
    // on ONE XCD (their slabs meet in its L2), every workgroup has exactly one unit and the grid is <= the CU count (launcher), so
    // all units are resident at once and the slices' mutual waits cannot deadlock.
    const int S = (CONV == 3) ? pa.split : 1;
    int sp_slice = 0, sp_tile = 0;
    if constexpr (CONV == 3) {
        if (S > 1) {
            const int idx = (int)(blockIdx.x >> 3);
            sp_slice = idx % S;
            sp_tile = (idx / S) * 8 + (int)(blockIdx.x & 7);
            if (sp_tile >= pa.total_tiles)
                return;
        }
    }
    const int nk = (p.k / BK) / S;
    const int kt0 = sp_slice * nk; // first K-tile of this workgroup's range
    const int my_tiles = S > 1 ? 1 : (pa.total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total_kt = my_tiles * nk; // this workgroup's flat K-tile sequence
    const unsigned per_batch = (unsigned)p.tiles_m * p.tiles_n;

    // step s of this workgroup -> (batch, m0, n0). Linear tile ids keep the XCD of the workgroup (gridDim.x % 8 == 0
    // whenever a workgroup has more than one tile), then the grouped raster of gemm256.hip.
    auto decode = [&](int s, int &ib, int &m0, int &n0) {
        // (every division by multiply-high + one correction: with plain `/` and `%` on run-time values this was 3 emulated
        // divisions of ~45 instructions, three times per tile — A cursor, B cursor, tile loop)
        unsigned wg = S > 1 ? (unsigned)sp_tile : xcd_remap((unsigned)s * gridDim.x + blockIdx.x, (unsigned)pa.total_tiles);
        unsigned ibu, rest;
        udivmod_m(wg, per_batch, pa.per_batch_m, ibu, rest);
        ib = (int)ibu;
        constexpr int GROUP_M = 8;
        const unsigned per_group = GROUP_M * p.tiles_n;
        unsigned group, in_group;
        udivmod_m(rest, per_group, pa.per_group_m, group, in_group);
        const int first_m = group * GROUP_M;
        const int gsz = min(p.tiles_m - first_m, GROUP_M);
        // floor(2^32 / gsz) for gsz = 1 .. 8
        const unsigned gm = gsz == 8 ? 0x20000000u : gsz == 7 ? 0x24924924u : gsz == 6 ? 0x2aaaaaaau : gsz == 5 ? 0x33333333u
                          : gsz == 4 ? 0x40000000u : gsz == 3 ? 0x55555555u : gsz == 2 ? 0x80000000u : 0xffffffffu;
        unsigned qn, rm;
        udivmod_m(in_group, (unsigned)gsz, gm, qn, rm);
        m0 = (first_m + (int)rm) * BM;
        n0 = (int)qn * BN_;
    };

    const unsigned tab0 = (unsigned)(unsigned long)IROCM_LDS_PTR(smem) + LDS_BYTES + (TRACE ? kTraceBytes : 0);
    auto tab_read = [&](int s, int which) __attribute__((always_inline)) { // (asm: see gemm256_common.h on what hipcc does around LDS-DMA)
        unsigned v;
        asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(tab0 + (unsigned)s * 8u + (unsigned)which * 4u));
        return v;
    };
    const int tab_n = pa.tab_n;
    const unsigned bias0 = tab0 + kTabBytes;

    const long lda = A_KMAJOR ? p.a_rs : p.a_cs;
    const long ldb = CONV ? (long)p.cv_hw : (B_KMAJOR ? p.b_cs : p.b_rs);
    const long a_step = A_KMAJOR ? (long)BK * 2 : (long)BK * lda * 2; // bytes per K-tile (wave-uniform)
    const long b_step = B_KMAJOR ? (long)BK * 2 : (long)BK * ldb * 2;

    // ---- staging cursors: the next K-tile each operand will fetch, in flat order ------------------
    unsigned a_off[4], b_off[4];
    // variant 10: wave row 0 issues EVERY piece (its own and those of the wave four below), row 1 none — row 0's LOAD phases end 440-540
    // cycles before the barrier (they wait for row 1's slower COMPUTE phases), row 1's are what row 0's bursts wait for
    constexpr bool ROW0DMA = (CONV == 0) && IROCM_KV == 10;
    constexpr int NBW = ROW0DMA ? 2 * NB : NB; // loads a counted wait may leave in flight
    unsigned a_off2[ROW0DMA ? 4 : 1], b_off2[ROW0DMA ? 4 : 1];
    const char *a_base, *b_base;
    int a_s = 0, a_kt = 0, a_G = 0; // tile step, K-tile inside it, flat index
    int b_s = 0, b_kt = 0, b_G = 0;
    // tap mode: where inside (channel block, tap) each cursor stands. A: weights [tap][F][C] — tap a_t, channel block a_cb.
    // B: (b_r, b_s3) and the byte offset b_koff of that tap's tile relative to the pointwise tile (cv_b0 at tap (0, 0) of block 0).
    constexpr bool TAPS = CONV == 3;
    // (ONE tap counter per cursor and value selects only: with separate row / column counters bumped in two branches hipcc merged the
    // two increments into one store through a selected POINTER, which kept all three variables in scratch memory — and, loaded from
    // there, in vector registers under exec masks)
    auto b_delta = [&](int t) { // the step of B's byte offset from tap t to the next one (tap = 3 r + s)
        return t == 8 ? p.cv_dcb : (t == 2 ? p.cv_dr01 : (t == 5 ? p.cv_dr12 : ((t == 0 || t == 3 || t == 6) ? p.cv_ds01 : p.cv_ds12)));
    };
    int a_t = TAPS ? kt0 % 9 : 0, a_cb = TAPS ? kt0 / 9 : 0;
    int b_t = a_t, b_koff = 0;
    if constexpr (TAPS) {
        // (one channel block's nine steps add up to 64 channels: BK * hw * 2 bytes)
        b_koff = p.cv_b0 + a_cb * (BK * 2 * p.cv_hw);
        for (int u = 0; u < b_t; ++u)
            b_koff += b_delta(u);
    }
    auto set_a_at = [&](int ib, int m0) __attribute__((always_inline)) {
        a_base = (const char *)((const unsigned short *)p.a + (long)ib * p.a_bs);
        if constexpr (CONV == 3) offs_k_rot(a_off, lda, m0, p.m, w, lane, sp_slice * (8 / S));
        else if constexpr (A_KMAJOR) offs_k(a_off, lda, m0, p.m, w, lane);
        else offs_mn(a_off, lda, m0, p.m, w, lane);
        if constexpr (ROW0DMA) {
            if constexpr (A_KMAJOR) offs_k(a_off2, lda, m0, p.m, w + 4, lane);
            else offs_mn(a_off2, lda, m0, p.m, w + 4, lane);
        }
    };
    auto set_b_at = [&](int ib, int n0) __attribute__((always_inline)) {
        b_base = (const char *)((const unsigned short *)p.b + (long)ib * p.b_bs);
        if constexpr (CONV) offs_mn_conv(b_off, p.cv_hw, p.cv_hwp, p.cv_hwp_m, (long)(CONV == 3 ? p.k / 9 : p.k) * p.cv_hw, n0, p.n, w, lane);
        else if constexpr (B_KMAJOR) offs_k_n<NT>(b_off, ldb, n0, p.n, w, lane);
        else offs_mn(b_off, ldb, n0, p.n, w, lane);
        if constexpr (ROW0DMA) {
            if constexpr (B_KMAJOR) offs_k_n<NT>(b_off2, ldb, n0, p.n, w + 4, lane);
            else offs_mn(b_off2, ldb, n0, p.n, w + 4, lane);
        }
    };
    auto set_a_tile = [&](int s) {
        int ib, m0, n0;
        decode(s, ib, m0, n0);
        set_a_at(ib, m0);
    };
    auto set_b_tile = [&](int s) {
        int ib, m0, n0;
        decode(s, ib, m0, n0);
        set_b_at(ib, n0);
    };
    // A cursor that crosses into tile s < kTileTab only REQUESTS the table entry (one ds_read in front of the phase's lgkmcnt wait);
    // finish_cursors, called behind the MFMA burst of the next C1 phase, turns it into base pointer and lane offsets —
    // before the cursor's next DMA (A: the next K-tile's L1, B: its L2). DIRECT (the prologue, before the table exists) and
    // steps beyond the table decode in place.
    // Both are finished in C1 and nowhere else: of the four barrier intervals of a K-tile only the one where this wave row's C1
    // faces the partner row's (long) L1 has slack behind the MFMA burst; a check behind C2 (which faces the short L2) cost the
    // 64-K-tile headline 2 %. B crosses in an L2 and waits for the C1 of the next K-tile (its next DMA is that K-tile's L2).
    int pend = 0; // bit 0: A, bit 1: B
    unsigned a_ent = 0u, b_ent = 0u;
    auto stage_a_next = [&](int buf, auto directc, auto issuec) __attribute__((always_inline)) { // issuec false: advance the cursor only
        constexpr bool ISSUE = decltype(issuec)::value;
        if constexpr (!ISSUE) {
        } else if constexpr (TAPS) {
            stage4(a_base + ((long)a_t * p.cv_atap + (long)a_cb * (BK * 2)), a_off, smem + buf * BUF_BYTES, w);
        } else {
            if constexpr (ROW0DMA) {
                if (wr == 0) {
                    stage4(a_base + (long)a_kt * a_step, a_off, smem + buf * BUF_BYTES, w);
                    stage4(a_base + (long)a_kt * a_step, a_off2, smem + buf * BUF_BYTES, w + 4);
                }
            } else if (!(TRACE && (pa.trace_fine == 2 || pa.trace_fine == 3))) // (experiment: trace_fine 2 = no A DMA, 3 = no DMA at all — timing only)
                stage4(a_base + (long)a_kt * a_step, a_off, smem + buf * BUF_BYTES, w);
        }
        if constexpr (TAPS) {
            a_cb += a_t == 8 ? 1 : 0;
            a_t = a_t == 8 ? 0 : a_t + 1;
        }
        ++a_G;
        if (++a_kt == nk) {
            a_kt = 0;
            if constexpr (TAPS)
                a_cb = 0; // (a_t is 0 again: nk = 9 * channel blocks)
            if (++a_s < my_tiles) {
                if (decltype(directc)::value || a_s >= tab_n) {
                    set_a_tile(a_s);
                } else {
                    a_ent = tab_read(a_s, 0);
                    pend |= 1;
                }
            }
        }
    };
    auto stage_b_next = [&](int buf, auto directc, auto issuec) __attribute__((always_inline)) {
        constexpr bool ISSUE = decltype(issuec)::value;
        if constexpr (!ISSUE) {
        } else if constexpr (TAPS) {
            stage_n<NB>(b_base + (long)b_koff, b_off, smem + buf * BUF_BYTES + OPER_BYTES, w);
        } else {
            if constexpr (ROW0DMA) {
                if (wr == 0) {
                    stage_n<NB>(b_base + (long)b_kt * b_step, b_off, smem + buf * BUF_BYTES + OPER_BYTES, w);
                    stage_n<NB>(b_base + (long)b_kt * b_step, b_off2, smem + buf * BUF_BYTES + OPER_BYTES, w + 4);
                }
            } else if (!(TRACE && (pa.trace_fine == 3)))
                stage_n<NB>(b_base + (long)b_kt * b_step, b_off, smem + buf * BUF_BYTES + OPER_BYTES, w);
        }
        if constexpr (TAPS) {
            // the step from tap b_t to the next one (tap = 3 r + s): s 0 -> 1, s 1 -> 2, row ends, the block's last tap
            b_koff += b_delta(b_t);
            b_t = b_t == 8 ? 0 : b_t + 1;
        }
        ++b_G;
        if (++b_kt == nk) {
            b_kt = 0;
            if constexpr (TAPS)
                b_koff = p.cv_b0; // (b_t is 0 again)
            if (++b_s < my_tiles) {
                if (decltype(directc)::value || b_s >= tab_n) {
                    set_b_tile(b_s);
                } else {
                    b_ent = tab_read(b_s, 1);
                    pend |= 2;
                }
            }
        }
    };
    auto finish_cursors = [&]() __attribute__((always_inline)) {
        if (pend) { // (wave-uniform; the entries are read only from here on, i.e. behind the lgkmcnt waits that followed their ds_reads)
            if (pend & 1) {
                asm volatile("" : "+v"(a_ent));
                const unsigned e = (unsigned)__builtin_amdgcn_readfirstlane((int)a_ent);
                set_a_at((int)(e >> 16), (int)(e & 0xffffu) * BM);
            }
            if (pend & 2) {
                asm volatile("" : "+v"(b_ent));
                const unsigned e = (unsigned)__builtin_amdgcn_readfirstlane((int)b_ent);
                set_b_at((int)(e >> 16), (int)(e & 0xffffu) * BN_);
            }
            pend = 0;
        }
    };

    // The first K-tiles go out NOW: the fragment address arithmetic, the accumulator clear (128 v_mov per lane) and everything
    // else up to the first barrier then run under their memory round trip instead of in front of it.
    stamp();
    { // (ONE decode for both cursors: the scalar decode is ~1 k cycles and sits in front of the kernel's first memory request)
        int ib0, m00, n00;
        decode(0, ib0, m00, n00);
        set_a_at(ib0, m00);
        set_b_at(ib0, n00);
    }
    stage_b_next(0, std::true_type{}, std::true_type{});
    stage_a_next(0, std::true_type{}, std::true_type{});
    if (total_kt > 1)
        stage_b_next(1, std::true_type{}, std::true_type{});
    if constexpr ((CONV == 0) && IROCM_KV == 9) { // schedule 9: wave row 1's A cursor runs one K-tile further ahead
        if (wr == 1 && total_kt > 1)
            stage_a_next(1, std::true_type{}, std::true_type{});
    }
    // the tile table: thread s decodes step s (under the round trip of the loads above; published by the barrier in front of the loop)
    if (t < my_tiles && t < tab_n) {
        int ib, m0, n0;
        decode(t, ib, m0, n0);
        const u32x2_t e = {((unsigned)ib << 16) | (unsigned)(m0 / BM), ((unsigned)ib << 16) | (unsigned)(n0 / BN_)};
        asm volatile("ds_write_b64 %0, %1" ::"v"(tab0 + (unsigned)t * 8u), "v"(e) : "memory");
    }

    // ---- per-lane LDS read addresses (gemm256.hip) ------------------------------------------------
    const unsigned lds0 = (unsigned)(unsigned long)IROCM_LDS_PTR(smem);
    const unsigned kmaj_lane = (unsigned)(l15 * 128 + (((g4 ^ (l15 >> 1)) & 3) | (((l15 >> 1) >> 2) << 2)) * 16);
    unsigned a_k[2], b_k[2];
    unsigned a_mn[8], b_mn[NT];
    const int mnf = ((l15 >> 2) & 3) | ((g4 & 1) << 2);
    const unsigned mn_lane = (unsigned)((g4 * 8 + (l15 >> 2)) * 512 + (l15 & 1) * 8);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a_k[ks] = lds0 + wr * (128 * 128) + (kmaj_lane ^ (ks * 64));
        b_k[ks] = lds0 + OPER_BYTES + wc * (16 * NT * 128) + (kmaj_lane ^ (ks * 64));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c16 = ((l15 >> 1) & 1) | ((((wr * 8 + i) ^ mnf)) << 1);
        a_mn[i] = lds0 + mn_lane + c16 * 16;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int c16 = ((l15 >> 1) & 1) | ((((wc * NT + j) ^ mnf)) << 1);
        b_mn[j] = lds0 + OPER_BYTES + mn_lane + c16 * 16;
    }
    auto flip_buf = [&](int d) { // d = +-BUF_BYTES: move the read addresses to the other K-tile buffer
        if constexpr (A_KMAJOR) { a_k[0] += d; a_k[1] += d; }
        else {
#pragma unroll
            for (int i = 0; i < 8; ++i) a_mn[i] += d;
        }
        if constexpr (B_KMAJOR) { b_k[0] += d; b_k[1] += d; }
        else {
#pragma unroll
            for (int j = 0; j < NT; ++j) b_mn[j] += d;
        }
    };

    f32x4 acc[8][NT];

    using FA = Frag<A_KMAJOR>;
    using FB = Frag<B_KMAJOR>;
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
    // B sub-tile q: column tiles q*2 .. q*2 + CNT - 1
    auto read_b = [&](auto qc, auto cntc, FB(&bq)[2][2]) {
        constexpr int q = decltype(qc)::value, CNT = decltype(cntc)::value;
        sfor<CNT>([&](auto jc) {
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
    // ZERO: the first K-tile of a tile — the ks = 0 MFMA of every accumulator takes a literal zero as its C operand, so the
    // accumulators are never cleared (128 v_mov per lane = ~530 cycles per tile boundary in the timeline, behind the store steps).
    auto compute = [&](auto qac, auto qbc, auto cntc, auto zeroc, FA(&aq)[4][2], FB(&bq)[2][2]) {
        constexpr int qa = decltype(qac)::value, qb = decltype(qbc)::value, CNT = decltype(cntc)::value;
        constexpr bool ZERO = decltype(zeroc)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < CNT; ++j)
                    acc[qa * 4 + i][qb * 2 + j] = Tr::mfma(bq[j][ks].get(), aq[i][ks].get(),
                                                           (ZERO && ks == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[qa * 4 + i][qb * 2 + j]);
    };

    // one k-step (ks) of compute(): 4 CNT MFMAs — the schedule-9 bursts put an LDS-DMA piece between such groups
    auto compute_ks = [&](auto qac, auto qbc, auto cntc, auto ksc, auto zeroc, FA(&aq)[4][2], FB(&bq)[2][2]) __attribute__((always_inline)) {
        constexpr int qa = decltype(qac)::value, qb = decltype(qbc)::value, CNT = decltype(cntc)::value, ks = decltype(ksc)::value;
        constexpr bool ZERO = decltype(zeroc)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < CNT; ++j)
                acc[qa * 4 + i][qb * 2 + j] = Tr::mfma(bq[j][ks].get(), aq[i][ks].get(),
                                                       (ZERO && ks == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[qa * 4 + i][qb * 2 + j]);
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using IJ1 = std::integral_constant<int, NJ1>;

    // ---- epilogue of one tile --------------------------------------------------------------------
    // ACT is the activation as a compile-time value (0 none, 1 relu, 5 Gelu with erf by Abramowitz-Stegun), one copy of the
    // epilogue each: apply_act's run-time switch, replicated for each of a lane's 128 accumulators, put ~1600 instructions
    // (erff, tanhf, expf bodies) between two stores — with the Gelu selected at run time BERT's FFN1 launch cost +85 us.
    // Sigmoid / tanh / erff-Gelu are served by the one-shot kernel (gemm.hip routes them; a run-time copy here beside the
    // three spills 528 bytes per lane).
    auto epilogue = [&](auto actc, int ib, int m0, int n0, int bslot) __attribute__((always_inline)) {
        constexpr int ACT = decltype(actc)::value;
        // Lane-derived values of the epilogue are rebuilt from an OPAQUE copy of the lane id (shadowing the kernel's lane / l15 / g4):
        // hipcc hoists loop-invariant lane arithmetic (store offsets, exchange sources, bias addresses) out of the tile loop, where it
        // then occupies VGPRs across the K loop — registers the 256-column builds do not have (two of them = 8 bytes of scratch).
        int lane_o = (int)(threadIdx.x & 63);
        asm volatile("" : "+v"(lane_o));
        const int lane = lane_o, l15 = lane_o & 15, g4 = lane_o >> 4;
        auto act1 = [&](float v) {
            if constexpr (ACT == 1) return v > 0.f ? v : 0.f;
            else if constexpr (ACT == 5) return gelu_poly(v);
            else return v;
        };
        unsigned short *C = (unsigned short *)p.c + (long)ib * p.c_bs;
        const unsigned short *bias = (const unsigned short *)p.bias;
        const bool interior = (m0 + BM <= p.m) && (n0 + BN_ <= p.n) && (p.n % 4 == 0);
        // Bias: none or ONE row vector [n] (bias_m == 0, bias_n == 1; launch_p admits nothing else): a lane's 4 NT values are
        // fetched ONCE per tile (one 8-byte load per column tile) instead of once per accumulator element (128 scalar loads
        // per lane, each followed by the compiler's vmcnt(0) — that also drained the DMA pipeline of the next tile). The
        // general forms (column vectors, full matrices) live in the one-shot kernel: their strides cost this kernel four
        // scalar registers it does not have — at 106 SGPRs ONE more 64-bit argument (the output batch stride) tipped the
        // 256-column build into 408-468 bytes of scratch per lane and 76 -> 116 us on BERT's FFN1.
        stamp(); // (TRACE: epilogue entry)
        const bool rowbias = bias != nullptr;
        float bv[NT][4];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                bv[j][r] = 0.f;
        if (rowbias) {
            const unsigned short *bb = bias + (long)ib * p.bias_b;
            if (n0 + BN_ <= p.n && ((((uintptr_t)(bb + n0)) & 15) == 0)) { // (the test of the tile loop's DMA: the values are in LDS)
                u32x2_t q[NT];
                const unsigned src = bias0 + (unsigned)bslot * kBiasSlotBytes + (unsigned)(wc * (16 * NT) + g4 * 4) * 2u;
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(q[j]) : "v"(src), "i"(j * 32));
                wait_lgkm0();
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    asm volatile("" : "+v"(q[j]));
                    bv[j][0] = Tr::to_f32((unsigned short)(q[j][0] & 0xffff));
                    bv[j][1] = Tr::to_f32((unsigned short)(q[j][0] >> 16));
                    bv[j][2] = Tr::to_f32((unsigned short)(q[j][1] & 0xffff));
                    bv[j][3] = Tr::to_f32((unsigned short)(q[j][1] >> 16));
                }
            } else {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int col = n0 + wc * (16 * NT) + j * 16 + g4 * 4;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        bv[j][r] = col + r < p.n ? Tr::to_f32(bb[col + r]) : 0.f;
                }
            }
        }
        stamp(); // (TRACE: bias)
        auto pack2 = [&](int i, auto jc, int row, unsigned (&pk)[2]) { // bias + activation + rounding of one 4-wide piece
            constexpr int j = decltype(jc)::value;
            const int col = n0 + wc * (16 * NT) + j * 16 + g4 * 4;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                v[r] = acc[i][j][r];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                v[r] = act1(v[r] + bv[j][r]);
            pk[0] = Tr::pack2(v[0], v[1]);
            pk[1] = Tr::pack2(v[2], v[3]);
        };
        if (interior && p.epi16 && (p.n % 8 == 0) && ((((uintptr_t)p.c) & 15) == 0)) {
            // 16-byte stores: lane groups g4 / g4^1 swap halves of a PAIR of column tiles (gemm256.hip); an odd last
            // column tile (NT = 3) goes out as 8-byte stores.
            // The epilogue is VALU-issue-bound, not store-bound (tools/probes/store_burst.hip: a CU retires this tile's 128 KiB in
            // 2-4 k cycles from 128 back-to-back dwordx4 stores; the epilogue took 4.6 k cycles of VALU issue per SIMD): per store
            // it now spends 8 bias adds, 4 packed conversions (v_cvt_pk_*: one per PAIR), 2 lane-group swaps and NO address
            // arithmetic — the row part of the address is wave-uniform (scalar base, bumped by scalar adds), the lane part a
            // 32-bit offset computed once per tile. (Before: one conversion per value + shift + or, and a 64-bit multiply-add
            // chain with a branch on the head-split layout per store: 36 VALU slots per store.)
            const bool odd = g4 & 1;
            stamp(); // (TRACE: 16-byte store path entered)
            // Head-split layout (the fused q / k / v projections: C[row][col] -> [row / S][col / D][row % S][col % D]) separates
            // the same way when S % 16 == 0: a wave's 16 rows of one store never straddle a sequence, so (row / S, row % S) of
            // the wave's first row is wave-uniform and advances by scalar adds; the lane adds l15 * D and its column part.
            const bool hs = p.hs_d != 0;
            if (!hs || (p.hs_s % 16 == 0 && (long)p.n * p.hs_s < (1l << 30))) {
                const unsigned ldc2 = (unsigned)p.n * 2u; // bytes per row of a plain C
                // Lane -> address map of a store. The texture addresser merges only CONSECUTIVE lanes into one request: with the
                // accumulator layout's natural map (lane = g4 * 16 + l15: sixteen consecutive lanes walk down sixteen rows) every
                // 16-byte lane is its own request and a CU retires 13-15 B/clk — the "store-issue-bound" 9-10 k-cycle epilogue of
                // round 2. With four consecutive lanes on one row (64 contiguous bytes per quad) the same stores retire at
                // 45-60 B/clk (tools/probes/store_burst2.hip: 128 KiB per CU in 2.3-2.6 k cycles instead of 7.4-8.2 k). The data
                // moves to that map by one ds_bpermute_b32 per dword (the LDS crossbar, no LDS memory): destination lane
                // L = 4 * row + chunk takes the registers of source lane 16 * g4 + row, g4 = 2 * (chunk & 1) + (chunk >> 1).
                const int qrow = lane >> 2, qchunk = lane & 3;
                const int bperm_src = (16 * (2 * (qchunk & 1) + (qchunk >> 1)) + qrow) * 4;
                const int bperm_src_odd = (16 * qchunk + qrow) * 4; // odd last tile: 8 bytes per lane, chunk = g4
                const unsigned lane_row = (unsigned)qrow * (hs ? (unsigned)p.hs_d * 2u : ldc2);
                auto col_part = [&](int col) -> unsigned { // byte offset contributed by the column
                    if (!hs)
                        return (unsigned)col * 2u;
                    // (the divisor is made opaque HERE: hipcc otherwise hoists the reciprocal of hs_d into the kernel prologue and
                    // keeps it in a VGPR for the whole kernel — the one register the 256-column builds spilled, reloaded in this
                    // epilogue behind a vmcnt(0) that also drains the next tile's operand DMAs)
                    unsigned hsd = (unsigned)p.hs_d;
                    asm volatile("" : "+s"(hsd));
                    const unsigned h = (unsigned)col / hsd, d = (unsigned)col - h * hsd;
                    return (h * (unsigned)p.hs_s * hsd + d) * 2u;
                };
                unsigned voff[NT / 2 > 0 ? NT / 2 : 1];
                sfor<NT / 2>([&](auto jpc) {
                    constexpr int jp = decltype(jpc)::value;
                    voff[jp] = lane_row + col_part(n0 + wc * (16 * NT) + jp * 32 + qchunk * 8);
                });
                unsigned voff_odd = 0;
                if constexpr (NT % 2 == 1)
                    voff_odd = lane_row + col_part(n0 + wc * (16 * NT) + (NT - 1) * 16 + qchunk * 4);
                // wave-uniform part: the wave's first row R0 = m0 + wr * 128 (+ 16 per step)
                const int R0 = m0 + wr * 128;
                int s0 = 0;              // R0 % S (head-split only)
                long step = 16l * ldc2;  // bytes between steps
                long wrap = 0;           // added when s0 wraps into the next sequence
                char *sbase;
                if (hs) {
                    const int bq = R0 / p.hs_s;
                    s0 = R0 - bq * p.hs_s;
                    sbase = (char *)C + ((long)bq * p.n * p.hs_s + (long)s0 * p.hs_d) * 2;
                    step = 32l * p.hs_d;
                    wrap = ((long)p.n - p.hs_d) * p.hs_s * 2;
                } else {
                    sbase = (char *)C + (long)R0 * (long)ldc2;
                }
                // One step = the wave's 16 rows x 64 NT columns: NT / 2 16-byte stores (+ one 8-byte store for an odd NT). The
                // lane exchange of step i + 1 is issued BEFORE the stores of step i, so that a store never waits out the
                // ds_bpermute latency of its own data (LDS results return in order: the wait is a counted lgkmcnt).
                u32x4_t ov[2][NT / 2 > 0 ? NT / 2 : 1];
                u32x2_t ov_odd[2];
                auto make = [&](int i, int slot) __attribute__((always_inline)) {
                    sfor<NT / 2>([&](auto jpc) {
                        constexpr int jp = decltype(jpc)::value;
                        unsigned pk[2][2];
                        pack2(i, std::integral_constant<int, jp * 2>{}, 0, pk[0]);
                        pack2(i, std::integral_constant<int, jp * 2 + 1>{}, 0, pk[1]);
                        // v_permlane16_swap: odd 16-lane rows of the first operand <-> even rows of the second: even lanes end
                        // with {own, right neighbour's} 4 + 4 columns of tile 2jp, odd lanes with {left neighbour's, own} of
                        // tile 2jp + 1 — one VALU op per dword, no LDS round trip, no selects
                        const u32x2_t t0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                        const u32x2_t t1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                        ov[slot][jp][0] = lds_bpermute(bperm_src, t0[0]);
                        ov[slot][jp][1] = lds_bpermute(bperm_src, t1[0]);
                        ov[slot][jp][2] = lds_bpermute(bperm_src, t0[1]);
                        ov[slot][jp][3] = lds_bpermute(bperm_src, t1[1]);
                    });
                    if constexpr (NT % 2 == 1) {
                        unsigned pk[2];
                        pack2(i, std::integral_constant<int, NT - 1>{}, 0, pk);
                        ov_odd[slot][0] = lds_bpermute(bperm_src_odd, pk[0]);
                        ov_odd[slot][1] = lds_bpermute(bperm_src_odd, pk[1]);
                    }
                };
                constexpr int kPerStep = (NT / 2) * 4 + (NT % 2) * 2; // exchanges in flight per step
                stamp(); // (TRACE: bias + address set-up)
                make(0, 0);
                stamp(); // (TRACE: the first pack + exchange)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    fence_sched();
                    if (i + 1 < 8) {
                        make(i + 1, (i + 1) & 1);
                        wait_lgkm<kPerStep>(); // step i's exchange has landed; step i + 1's stays in flight under the stores
                    } else {
                        wait_lgkm<0>();
                    }
                    sfor<NT / 2>([&](auto jpc) {
                        constexpr int jp = decltype(jpc)::value;
                        *(u32x4_t *)(sbase + voff[jp]) = ov[i & 1][jp];
                    });
                    if constexpr (NT % 2 == 1)
                        *(u32x2_t *)(sbase + voff_odd) = ov_odd[i & 1];
                    fence_sched();
                    stamp(); // (TRACE build: progress of the store sequence)
                    sbase += step;
                    if (hs) {
                        s0 += 16;
                        if (s0 >= p.hs_s) {
                            s0 -= p.hs_s;
                            sbase += wrap;
                        }
                    }
                }
                return;
            }
            // head-split layout with a sequence length that is not a multiple of 16: per-store c_off (divisions)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = m0 + wr * 128 + i * 16 + l15;
                sfor<NT / 2>([&](auto jpc) {
                    constexpr int jp = decltype(jpc)::value;
                    unsigned pk[2][2];
                    pack2(i, std::integral_constant<int, jp * 2>{}, row, pk[0]);
                    pack2(i, std::integral_constant<int, jp * 2 + 1>{}, row, pk[1]);
                    // v_permlane16_swap: odd 16-lane rows of the first operand <-> even rows of the second: even lanes end
                    // with {own, right neighbour's} 4 + 4 columns of tile 2jp, odd lanes with {left neighbour's, own} of
                    // tile 2jp + 1 — one VALU op per dword, no LDS round trip, no selects
                    const u32x2_t t0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                    const u32x2_t t1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                    u32x4_t o;
                    o[0] = t0[0]; o[1] = t1[0]; o[2] = t0[1]; o[3] = t1[1];
                    const int col = n0 + wc * (16 * NT) + (jp * 2 + (odd ? 1 : 0)) * 16 + (g4 & ~1) * 4;
                    *(u32x4_t *)(C + c_off(p, row, col)) = o;
                });
                if constexpr (NT % 2 == 1) {
                    unsigned pk[2];
                    pack2(i, std::integral_constant<int, NT - 1>{}, row, pk);
                    const int col = n0 + wc * (16 * NT) + (NT - 1) * 16 + g4 * 4;
                    u32x2_t o2;
                    o2[0] = pk[0]; o2[1] = pk[1];
                    *(u32x2_t *)(C + c_off(p, row, col)) = o2;
                }
            }
            return;
        }
        if (interior) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = m0 + wr * 128 + i * 16 + l15;
                sfor<NT>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    unsigned pk[2];
                    pack2(i, jc, row, pk);
                    const int col = n0 + wc * (16 * NT) + j * 16 + g4 * 4;
                    u32x2_t o2;
                    o2[0] = pk[0]; o2[1] = pk[1];
                    *(u32x2_t *)(C + c_off(p, row, col)) = o2;
                });
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = m0 + wr * 128 + i * 16 + l15;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = n0 + wc * (16 * NT) + j * 16 + g4 * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (row < p.m && col + r < p.n) {
                        C[c_off(p, row, col + r)] = Tr::from_f32(act1(acc[i][j][r] + bv[j][r]));
                    }
                }
            }
        }
    };

    // conv mode: the per-filter bias of the lane's eight accumulator rows lives in registers ACROSS tiles and is (re)loaded only
    // when the tile's filter block m0 changes (layers with F <= 256 have one block: loaded once per kernel), before the tile's K
    // loop so that nothing waits for it at the head of the epilogue. (Not in the 256-column residual copy: it has no registers.)
    constexpr bool kBiasEarly = (CONV == 1) || (CONV == 3) || (CONV == 2 && NT < 4);
    float cbias[kBiasEarly ? 8 : 1];
    int cbias_m0 = -1;
    auto load_cbias = [&](int m0) __attribute__((always_inline)) {
        if constexpr (kBiasEarly) {
            if (m0 != cbias_m0) { // (wave-uniform)
                cbias_m0 = m0;
                const unsigned short *bias = (const unsigned short *)p.bias;
                if (bias != nullptr) {
                    unsigned short braw[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int row = m0 + wr * 128 + i * 16 + l15;
                        braw[i] = bias[row < p.m ? row : p.m - 1];
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        cbias[i] = Tr::to_f32(braw[i]);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        cbias[i] = 0.f;
                }
            }
        }
    };

    // ---- conv-mode epilogue: rows are filters, columns pixel slots; Y is NCHW ---------------------------------------
    auto epilogue_conv = [&](auto actc, auto resc, int m0, int n0, unsigned rbmask) __attribute__((always_inline)) {
        constexpr int ACT = decltype(actc)::value;
        // (opaque lane id: see epilogue)
        int lane_o = (int)(threadIdx.x & 63);
        asm volatile("" : "+v"(lane_o));
        const int lane = lane_o, l15 = lane_o & 15, g4 = lane_o >> 4;
        constexpr bool has_res = decltype(resc)::value; // compile-time: the residual registers exist in the residual copies only
        auto act1 = [&](float v) {
            if constexpr (ACT == 1) return v > 0.f ? v : 0.f;
            else return v;
        };
        unsigned short *Y = (unsigned short *)p.c;
        const unsigned short *bias = (const unsigned short *)p.bias;
        const int HW = p.cv_hw, HWP = p.cv_hwp, F = p.m;
        const bool odd = g4 & 1;
        // this lane's column pieces: NT / 2 runs of 8 slots (after the lane-group exchange) and, for odd NT, one run of 4
        // Addresses as in the GEMM epilogue: the filter row of a store is (wave-uniform first row) + l15, so the row part of the
        // address is a scalar base bumped by scalar adds and the lane keeps ONE 32-bit byte offset per column run
        // (the launcher admits < 2^31 elements): no vector address arithmetic per store.
        // Stores use the quad-contiguous lane map of the GEMM epilogue (four consecutive lanes = 32 consecutive pixel slots of one
        // filter row; data moved by ds_bpermute_b32): lane L owns filter row qrow = L / 4 and the 8-slot run qchunk = L % 4.
        const int qrow = lane >> 2, qchunk = lane & 3;
        const int bperm_src = (16 * (2 * (qchunk & 1) + (qchunk >> 1)) + qrow) * 4;
        const int bperm_src_odd = (16 * qchunk + qrow) * 4;
        unsigned pbase[NT / 2 > 0 ? NT / 2 : 1]; // byte offset of (img, filter qrow, pix)
        int plive[NT / 2 > 0 ? NT / 2 : 1];       // live pixels of the run: 8 inside a plane, fewer at its ragged end, 0 past the tensor
        sfor<NT / 2>([&](auto jpc) {
            constexpr int jp = decltype(jpc)::value;
            const int col = n0 + wc * (16 * NT) + jp * 32 + qchunk * 8;
            unsigned imgu, pixu;
            udivmod_m((unsigned)col, (unsigned)HWP, p.cv_hwp_m, imgu, pixu);
            const int img = (int)imgu, pix = (int)pixu;
            pbase[jp] = (unsigned)((img * F + qrow) * HW + pix) * 2u;
            plive[jp] = col < p.n ? min(8, HW - pix) : 0;
        });
        unsigned obase = 0;
        int olive = 0;
        if constexpr (NT % 2 == 1) {
            const int col = n0 + wc * (16 * NT) + (NT - 1) * 16 + qchunk * 4;
            unsigned imgu, pixu;
            udivmod_m((unsigned)col, (unsigned)HWP, p.cv_hwp_m, imgu, pixu);
            const int img = (int)imgu, pix = (int)pixu;
            obase = (unsigned)((img * F + qrow) * HW + pix) * 2u;
            olive = col < p.n ? max(0, min(4, HW - pix)) : 0;
        }
        const int R0 = m0 + wr * 128;                      // wave-uniform first filter row
        char *sbase = (char *)Y + (long)R0 * HW * 2;
        const long sstep = 32l * HW;                        // 16 filter rows
        const bool rows_inside = m0 + BM <= F;              // no filter-row check needed (uniform)
        // per-filter bias of the lane's eight accumulator rows: eight UNCONDITIONAL loads (row clamped; a missing bias is a uniform
        // branch around all of them) and one wait. (As `cond ? bias[row] : 0` each load sat in its own branch with an
        // s_waitcnt vmcnt(0) behind it: eight serialised memory round trips at the head of every tile's epilogue.)
        float bvr[8];
        if constexpr (kBiasEarly) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                bvr[i] = cbias[i];
        } else if (bias != nullptr) {
            unsigned short braw[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = m0 + wr * 128 + i * 16 + l15;
                braw[i] = bias[row < F ? row : F - 1];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                bvr[i] = Tr::to_f32(braw[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                bvr[i] = 0.f;
        }
        // Residual: fetched in the STORE layout (the lane's 8-slot run of filter row qrow — four consecutive lanes read 64
        // contiguous bytes, the map the texture addresser merges; in the accumulator layout's own map these loads crawled like
        // the stores did: the residual cost a C64 -> F256 56 x 56 layer 71 us on top of its 51) and added after the lane exchange
        // on the packed values: y = act(round(conv + bias) + residual) — the arithmetic of the reference's separate Conv and Add
        // kernels (and of conv_s1.hip's row-wise epilogue). Every run of the tile is fetched BEFORE the first store (the
        // operand-fragment registers are dead here) through a range-checked buffer descriptor (dead runs read zeros): a load
        // inside the store sequence would have to drain the stores in front of it (loads and stores share vmcnt).
        u32x4_t rq[has_res ? 8 : 1][has_res ? (NT / 2 > 0 ? NT / 2 : 1) : 1];
        u32x2_t rq_odd[has_res ? 8 : 1];
        // (Round 6: requesting the NEXT tile's runs from the current epilogue — 32 registers live across the K loop in the 128-column
        // copy, inline-asm loads the K loop's counted waits cover — was built, parity-green, and moved nothing: C64 -> F256 @56 x 56
        // 108.6 us with and without. A tile moves 176 KB in 18.5 k cycles = 9.5 B / clk / CU, the per-CU streaming rate of this chip
        // (MI355X_MICROARCH.md: ~10): these layers are bound by the mixed read / write bandwidth, not by the round trip. Removed.)
        if constexpr (has_res) {
            const long total_bytes = (long)p.cv_res_bytes;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.cv_res), 0, (int)p.cv_res_bytes, 0x00020000);
            const unsigned rowb = (unsigned)R0 * (unsigned)HW * 2u; // wave-uniform byte offset of the wave's first filter row
            auto fetch = [&](unsigned off, int live, auto nc) { // `live` leading 16-bit values of an N-value run at byte offset off
                constexpr int N = decltype(nc)::value;
                u32x4_t v = {0u, 0u, 0u, 0u};
                if (live <= 0)
                    return v;
                if (off + 2u * N <= (unsigned)total_bytes) {
                    if constexpr (N == 8) v = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
                    else {
                        const u32x2_t h = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0));
                        v[0] = h[0]; v[1] = h[1];
                    }
                } else {
                    // the ragged run at the very end of the tensor: the descriptor's range check works per (misaligned) dword
                    // and would zero the last live element together with the bytes behind it — fetch by element
#pragma unroll
                    for (int q = 0; q < N; ++q) {
                        const unsigned e = q < live ? (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)off + 2 * q, 0, 0) : 0u;
                        v[q >> 1] |= e << ((q & 1) * 16);
                    }
                }
                return v;
            };
            // Only a tile that holds the LAST run of the tensor on a plane of hw % 8 != 0 pixels needs the element-wise form
            // (wave-uniform test). Everywhere else the fetch is straight-line code: a dead run gets an out-of-range offset and reads
            // zeros. (With the per-run branches of `fetch` hipcc put an s_waitcnt vmcnt(0) behind EVERY load — 16-24 serialised
            // memory round trips per tile: the residual cost a C64 -> F256 56 x 56 layer 67 us for 205 MB.)
            const bool tail_tile = (HW % 8 != 0) && (n0 + BN_ + HWP > p.n) && (m0 + BM >= F);
            if (tail_tile) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bool rowok = rows_inside || (R0 + i * 16 + qrow < F);
                    sfor<NT / 2>([&](auto jpc) {
                        constexpr int jp = decltype(jpc)::value;
                        rq[i][jp] = fetch(rowb + (unsigned)i * (unsigned)sstep + pbase[jp], rowok ? plive[jp] : 0, std::integral_constant<int, 8>{});
                    });
                    if constexpr (NT % 2 == 1) {
                        const u32x4_t v = fetch(rowb + (unsigned)i * (unsigned)sstep + obase, rowok ? olive : 0, std::integral_constant<int, 4>{});
                        rq_odd[i][0] = v[0]; rq_odd[i][1] = v[1];
                    }
                }
            } else {
                const unsigned dead = (unsigned)total_bytes; // past the descriptor's range: zeros, no memory access
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bool rowok = rows_inside || (R0 + i * 16 + qrow < F);
                    sfor<NT / 2>([&](auto jpc) {
                        constexpr int jp = decltype(jpc)::value;
                        const unsigned off = (rowok && plive[jp] > 0) ? rowb + (unsigned)i * (unsigned)sstep + pbase[jp] : dead;
                        rq[i][jp] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
                    });
                    if constexpr (NT % 2 == 1) {
                        const unsigned off = (rowok && olive > 0) ? rowb + (unsigned)i * (unsigned)sstep + obase : dead;
                        rq_odd[i] = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0));
                    }
                }
            }
        }
        auto add_res = [&](unsigned o, unsigned r) -> unsigned { // two packed values + two residual values, activation, rounding
            float lo = Tr::to_f32((unsigned short)(o & 0xffffu)) + Tr::to_f32((unsigned short)(r & 0xffffu));
            float hi = Tr::to_f32((unsigned short)(o >> 16)) + Tr::to_f32((unsigned short)(r >> 16));
            return Tr::pack2(act1(lo), act1(hi));
        };
        auto pack4 = [&](int i, auto jc, unsigned (&pk)[2]) { // bias (+ activation when there is no residual) + rounding of the lane's 4 slots of tile j
            constexpr int j = decltype(jc)::value;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[i][j][r] + bvr[i];
                if constexpr (!has_res)
                    v[r] = act1(v[r]);
            }
            pk[0] = Tr::pack2(v[0], v[1]);
            pk[1] = Tr::pack2(v[2], v[3]);
        };
        auto store_run = [&](char *dstc, const u32x4_t &o, int live) { // `live` of the 8 values in o
            if (live == 8) {
                *(u32x4_t *)dstc = o; // (8-byte aligned when the plane is not a multiple of 8 pixels: dwordx4 only needs dwords)
                return;
            }
            unsigned short *dst = (unsigned short *)dstc;
            int e = 0;
            if (live & 4) {
                u32x2_t v;
                v[0] = o[0]; v[1] = o[1];
                *(u32x2_t *)dst = v;
                e = 4;
            }
            if (live & 2) {
                *(unsigned *)(dst + e) = e ? o[2] : o[0];
                e += 2;
            }
            if (live & 1)
                dst[e] = (unsigned short)((e == 0 ? o[0] : e == 2 ? o[1] : e == 4 ? o[2] : o[3]) & 0xffff);
        };
        // (as in the GEMM epilogue: the exchange of step i + 1 is in flight under the stores of step i)
        u32x4_t ov[2][NT / 2 > 0 ? NT / 2 : 1];
        unsigned ov_odd[2][2];
        auto make = [&](int i, int slot) __attribute__((always_inline)) {
            sfor<NT / 2>([&](auto jpc) {
                constexpr int jp = decltype(jpc)::value;
                unsigned pk[2][2];
                pack4(i, std::integral_constant<int, jp * 2>{}, pk[0]);
                pack4(i, std::integral_constant<int, jp * 2 + 1>{}, pk[1]);
                // lane groups g4 / g4 ^ 1 trade halves (v_permlane16_swap, as in the GEMM epilogue): an even group ends with 8
                // consecutive slots of tile 2 jp, an odd one with 8 of tile 2 jp + 1
                const u32x2_t t0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                const u32x2_t t1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                ov[slot][jp][0] = lds_bpermute(bperm_src, t0[0]);
                ov[slot][jp][1] = lds_bpermute(bperm_src, t1[0]);
                ov[slot][jp][2] = lds_bpermute(bperm_src, t0[1]);
                ov[slot][jp][3] = lds_bpermute(bperm_src, t1[1]);
            });
            if constexpr (NT % 2 == 1) {
                unsigned pk[2];
                pack4(i, std::integral_constant<int, NT - 1>{}, pk);
                ov_odd[slot][0] = lds_bpermute(bperm_src_odd, pk[0]);
                ov_odd[slot][1] = lds_bpermute(bperm_src_odd, pk[1]);
            }
        };
        constexpr int kPerStep = (NT / 2) * 4 + (NT % 2) * 2;
        // Two copies of the store sequence, chosen by ONE uniform branch: planes of hw % 8 == 0 pixels (56 x 56, 28 x 28) have
        // no ragged runs — a run is whole or dead — so their copy stores 16 (8) bytes under one lane predicate; the general copy
        // keeps the 4 + 2 + 1 element stores of a plane's last run (ten exec-mask branches per 16-row step when inlined for all).
        auto store_all = [&](auto fullc) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(fullc)::value;
            make(0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int srow = m0 + wr * 128 + i * 16 + qrow; // the filter row this lane STORES (after the lane exchange)
                fence_sched();
                if (i + 1 < 8) {
                    make(i + 1, (i + 1) & 1);
                    wait_lgkm<kPerStep>();
                } else {
                    wait_lgkm<0>();
                }
                const bool rowok = (rows_inside || srow < F) && ((rbmask >> i) & 1u); // (split-K: the row blocks this slice finishes)
                sfor<NT / 2>([&](auto jpc) {
                    constexpr int jp = decltype(jpc)::value;
                    u32x4_t o = ov[i & 1][jp];
                    if constexpr (has_res) {
#pragma unroll
                        for (int d = 0; d < 4; ++d)
                            o[d] = add_res(o[d], rq[i][jp][d]);
                    }
                    if constexpr (FULL) {
                        if (rowok && plive[jp] > 0)
                            *(u32x4_t *)(sbase + pbase[jp]) = o;
                    } else {
                        if (rowok && plive[jp] > 0)
                            store_run(sbase + pbase[jp], o, plive[jp]);
                    }
                });
                if constexpr (NT % 2 == 1) {
                    unsigned pk0 = ov_odd[i & 1][0], pk1 = ov_odd[i & 1][1];
                    if constexpr (has_res) {
                        pk0 = add_res(pk0, rq_odd[i][0]);
                        pk1 = add_res(pk1, rq_odd[i][1]);
                    }
                    if (rowok && olive > 0) {
                        unsigned short *dst = (unsigned short *)(sbase + obase);
                        if (FULL || olive == 4) {
                            u32x2_t v;
                            v[0] = pk0; v[1] = pk1;
                            *(u32x2_t *)dst = v;
                        } else {
                            if (olive & 2)
                                *(unsigned *)dst = pk0;
                            if (olive & 1)
                                dst[olive & 2] = (unsigned short)((olive & 2 ? pk1 : pk0) & 0xffff);
                        }
                    }
                }
                fence_sched();
                sbase += sstep;
            }
        };
        if constexpr (has_res && NT == 4) {
            store_all(std::false_type{}); // (the 256-column residual copy is out of registers: one copy of the sequence only)
        } else {
            if (HW % 8 == 0) store_all(std::true_type{});
            else store_all(std::false_type{});
        }
    };

    // ---- split-K exchange (tap mode) ----------------------------------------------------------------------------------
    // After its K range every wave holds PARTIAL sums of its 128 x 16 NT block. Slice d of the tile finishes row blocks
    // [d * 8 / S, (d + 1) * 8 / S) of EVERY wave: a wave hands the other slices' row blocks to the wave of the same index in those
    // slices (fragment layout, 1 KiB per (row block, column tile): fully coalesced 16-byte stores) and adds what they hand it.
    // All eight waves of all S workgroups move data at once — S - 1 of S of a wave's accumulators out, as many in — and each
    // workgroup then stores 1 / S of the tile. Hand-off as MI355X_MICROARCH.md / cdna_hip_programming.md Guideline 16, form R1, per
    // WAVE PAIR (no workgroup barrier): payload written through (sc1) -> the storing wave drains vmcnt -> ONE flag word per (source,
    // destination, wave) stored at agent scope -> the consumer polls that word relaxed (bounded) -> payload read with sc1 loads ->
    // the consumer zeroes the flag (every flag has exactly one consumer: the words are zero again when the kernel ends; the launcher
    // zeroes them once at allocation). Sums: own partial first, then the other slices' in ascending slice order — fixed per output
    // element, so results are reproducible run to run.
    // Which row blocks a slice keeps must not select REGISTERS at run time (that would put the accumulators into scratch), and one
    // code copy per (factor, slice) made hipcc spill ~350 registers. Instead the A tile is staged with its 16-row blocks ROTATED by
    // rot = slice * (8 / S) (set_a_at: LDS row block b of a wave row holds filter block (b + rot) % 8 — per-lane DMA offsets, free),
    // so in EVERY slice the blocks it keeps are accumulators 0 .. 8 / S - 1 and the ones it hands out are the rest; the slab is indexed
    // by the PHYSICAL block (b + rot) % 8, which is scalar arithmetic. All loads of one source are issued back to back into their own
    // registers (<= 8 pieces = 32 registers at a time) and added behind counted waits (left to itself hipcc reused one temporary and
    // waited vmcnt(0) behind every load: sixteen serialised round trips, 5 us).
    auto exchange = [&]() __attribute__((always_inline)) -> unsigned {
        int lane_o = (int)(threadIdx.x & 63);
        asm volatile("" : "+v"(lane_o));
        const int per = 8 / S, rot = sp_slice * per;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pa.slab, 0, (int)pa.slab_bytes, 0x00020000);
        // byte offset of (source slice, physical row block 0, column tile 0) of this wave and lane; a (row block, column tile) piece is 1 KiB
        auto slab_at = [&](int src) -> int {
            return (int)(((((unsigned)(sp_tile * S + src) * 8u + (unsigned)w) * 8u) * (unsigned)NT * 64u + (unsigned)lane_o) * 16u);
        };
        auto flag_of = [&](int src, int dst) -> unsigned * { return pa.flags + (((sp_tile * S + src) * S + dst) * 8 + w); };
        stamp(); // (TRACE: exchange entry)
        const int mine = slab_at(sp_slice);
#pragma unroll
        for (int a = 2; a < 8; ++a) { // (accumulators 0 and 1 are kept under every factor)
            if (a >= per) {           // (wave-uniform)
                const int pb = (a + rot) & 7;
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[a][j]), rs, mine + (pb * NT + j) * 1024, 0, 16);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (this wave's payload is written through before its flags go out)
        stamp(); // (TRACE: payload drained)
        if (lane_o == 0) {
            for (int d = 0; d < S; ++d)
                if (d != sp_slice)
                    __hip_atomic_store(flag_of(sp_slice, d), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        auto take = [&](auto perc, int src) __attribute__((always_inline)) {
            constexpr int PER = decltype(perc)::value, PIECES = PER * NT, BATCH = PIECES < 8 ? PIECES : 8;
            const int theirs = slab_at(src) + rot * NT * 1024; // my physical blocks rot .. rot + PER - 1 in the source's slab
            sfor<PIECES / BATCH>([&](auto bc) {
                constexpr int b0 = decltype(bc)::value * BATCH;
                u32x4_t in[BATCH];
                sfor<BATCH>([&](auto qc) {
                    constexpr int q = b0 + decltype(qc)::value;
                    in[q - b0] = __builtin_amdgcn_raw_buffer_load_b128(rs, theirs + q * 1024, 0, 16);
                });
                fence_sched();
                sfor<BATCH>([&](auto qc) {
                    constexpr int q = b0 + decltype(qc)::value;
                    acc[q / NT][q % NT] += __builtin_bit_cast(f32x4, in[q - b0]);
                });
                fence_sched();
            });
        };
        for (int src = 0; src < S; ++src) {
            if (src == sp_slice)
                continue;
            unsigned *fp = flag_of(src, sp_slice);
            const long long t0 = (long long)wall_clock64();
            while (__hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                __builtin_amdgcn_s_sleep(2);
                if ((long long)wall_clock64() - t0 > 200000000ll) { // 2 s at 100 MHz: a lost partner ends as wrong numbers, not as a hung GPU —
                    if (lane_o == 0 && pa.err)                     // and as an error at the next infini_rocm_runtime_sync (which re-zeroes the flags)
                        __hip_atomic_store(pa.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
            stamp(); // (TRACE: this source's flag seen)
            if (S == 2) take(std::integral_constant<int, 4>{}, src); // (the launcher admits 2 and 4)
            else take(std::integral_constant<int, 2>{}, src);
            if (lane_o == 0)
                __hip_atomic_store(fp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            stamp(); // (TRACE: this source's row blocks added)
        }
        return (1u << per) - 1u; // the epilogue's steps 0 .. per - 1 (filter rows m0 + rot * 16 ...: see the tile loop)
    };

    // ---- the flat K-tile pipeline ------------------------------------------------------------------
    FA aq[4][2];
    FB bq0[2][2], bq1[2][2];
    // tap mode: tv[j] holds, for this lane's slot of column tile j of the tile being accumulated, one validity bit per tap
    // (bit 3 r + s: the tap's input pixel lies inside the image); c_t = the tap of the K-tile being computed. A B fragment whose bit
    // is clear is zeroed — that IS the convolution's zero padding (what the moved run fetched instead is a neighbour's pixel).
    unsigned tv[TAPS ? NT : 1];
    int c_t = 0;
    auto mask_b = [&](auto qc, auto cntc, FB(&bq)[2][2]) __attribute__((always_inline)) {
        if constexpr (TAPS && !B_KMAJOR) {
            constexpr int q = decltype(qc)::value, CNT = decltype(cntc)::value;
            sfor<CNT>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int m = __builtin_amdgcn_sbfe((int)tv[q * 2 + j], (unsigned)c_t, 1u); // 0 or -1
                sfor<2>([&](auto kc) {
                    constexpr int ks = decltype(kc)::value;
                    // (the empty asm statements keep the AND a 32-bit operation: left to itself hipcc re-types it as an AND of
                    // four 16-bit elements and emits v_and_b32 + v_and_b32_sdwa per dword — twice the instructions)
                    typedef int i32x2_ __attribute__((ext_vector_type(2)));
                    i32x2_ lo = __builtin_bit_cast(i32x2_, bq[j][ks].lo), hi = __builtin_bit_cast(i32x2_, bq[j][ks].hi);
                    asm("" : "+v"(lo), "+v"(hi));
                    lo[0] &= m; lo[1] &= m;
                    hi[0] &= m; hi[1] &= m;
                    asm("" : "+v"(lo), "+v"(hi));
                    bq[j][ks].lo = __builtin_bit_cast(s16x4_t, lo);
                    bq[j][ks].hi = __builtin_bit_cast(s16x4_t, hi);
                });
            });
        }
    };
    constexpr int KV = (CONV == 0) ? IROCM_KV : 0;
    if constexpr (KV == 4 || KV == 5) {
        // variant 4: accumulators in AGPRs. hipcc selects the VGPR form of the MFMAs when a kernel is limited to 256 registers and
        // never mentions an AGPR; one empty asm with an "a" operand flips that (SIMachineFunctionInfo: mayUseAGPRs) — the question being
        // whether C / D traffic in the accumulator half of the file stops competing with the LDS returns for the arch-VGPR ports.
        int agpr_probe = 0;
        asm volatile("" : "+a"(agpr_probe));
    }
    auto fine_stamp = [&]() __attribute__((always_inline)) { // (TRACE, IROCM_GEMM_TRACE_FINE=4: two extra stamps inside each LOAD phase)
        if constexpr (TRACE) { if (pa.trace_fine == 4) stamp(); }
    };
    constexpr int SCHED = (KV == 4 || KV == 7 || KV == 8 || KV == 10) ? 0 : KV; // (variant 4 = schedule 0 with AGPR accumulators; 6 = NO DMA in the K loop: timing-only ablation)
    // Schedule 9 (round 6): NO LDS-DMA in the LOAD phases — every piece is issued by a wave in a COMPUTE phase, one piece behind each
    // group of 8 MFMAs, so that the 16 pieces of a phase no longer reach the texture addresser as one burst in front of the barrier the
    // partner row waits at (the no-DMA ablation, variant 6, runs 14-16 % faster than variant 0: that is what the bursts cost).
    // Which operand in which phase is fixed by when its LDS region is free (row 0 runs one interval ahead of row 1):
    //   row 0:  C1(t): A(t + 1) -> other buffer    C2(t): B(t + 2) -> this buffer, then the counted wait (A(t + 1), B(t + 1) landed)
    //   row 1:  C1(t): B(t + 2) -> this buffer     C2(t): A(t + 2) -> this buffer  (its A cursor runs one K-tile further ahead: the
    //           prologue issues its pieces of A(1)); its counted wait stays at the end of L2 (behind C1's B pieces).
    // A cursor that crosses into the next tile finishes in place (table entry + lgkm wait): the piece sequence sits behind MFMAs in flight.
    // Two GENERIC cursors, one per COMPUTE phase — X is staged in C1, Y in C2 — loaded BY VALUE from the A / B cursors behind the
    // prologue (row 0: X = A, Y = B; row 1: X = B, Y = A) and used alone from then on: every run-time "A or B" choice on the named
    // cursors (arrays, counters bumped in two branches) made hipcc keep them in scratch memory behind a selected pointer.
    const char *x_base = nullptr, *y_base = nullptr;
    unsigned x_off[4] = {0u, 0u, 0u, 0u}, y_off[4] = {0u, 0u, 0u, 0u};
    int x_kt = 0, x_s = 0, x_G = 0, y_kt = 0, y_s = 0, y_G = 0;
    long x_step = 0, y_step = 0;
    bool x_is_a = true;   // (y is the other operand)
    int x_np = 4, y_np = 4; // pieces per wave and K-tile
    int x_lds = 0, y_lds = 0; // byte offset of this wave's first piece inside a K-tile buffer
    auto cursor_set = [&](bool is_a, int ib, int m0, int n0, const char *&base, unsigned (&off)[4]) __attribute__((always_inline)) {
        if (is_a) { // (wave-uniform; both arms write the SAME variables)
            base = (const char *)((const unsigned short *)p.a + (long)ib * p.a_bs);
            if constexpr (A_KMAJOR) offs_k(off, lda, m0, p.m, w, lane);
            else offs_mn(off, lda, m0, p.m, w, lane);
        } else {
            base = (const char *)((const unsigned short *)p.b + (long)ib * p.b_bs);
            if constexpr (B_KMAJOR) offs_k_n<NT>(off, ldb, n0, p.n, w, lane);
            else offs_mn(off, ldb, n0, p.n, w, lane);
        }
    };
    auto cursor_cross = [&](bool is_a, int step_s, const char *&base, unsigned (&off)[4]) __attribute__((always_inline)) {
        int ib, m0, n0;
        if (step_s < tab_n) {
            unsigned e0 = tab_read(step_s, 0), e1 = tab_read(step_s, 1);
            wait_lgkm0();
            asm volatile("" : "+v"(e0), "+v"(e1));
            e0 = (unsigned)__builtin_amdgcn_readfirstlane((int)e0);
            e1 = (unsigned)__builtin_amdgcn_readfirstlane((int)e1);
            ib = (int)(e0 >> 16);
            m0 = (int)(e0 & 0xffffu) * BM;
            n0 = (int)(e1 & 0xffffu) * BN_;
        } else {
            decode(step_s, ib, m0, n0);
        }
        cursor_set(is_a, ib, m0, n0, base, off);
    };
    auto burst9 = [&](auto qac, auto zeroc, auto isxc, int dbuf) __attribute__((always_inline)) {
        constexpr int qa = decltype(qac)::value;
        constexpr bool ISX = decltype(isxc)::value; // the cursor of this phase: compile time (C1: X, C2: Y)
        const char *&base = ISX ? x_base : y_base;
        unsigned (&off)[4] = ISX ? x_off : y_off;
        int &kt = ISX ? x_kt : y_kt, &st = ISX ? x_s : y_s, &G_ = ISX ? x_G : y_G;
        const long step = ISX ? x_step : y_step;
        const int np = ISX ? x_np : y_np, lds = ISX ? x_lds : y_lds;
        const bool is_a = ISX ? x_is_a : !x_is_a;
        const bool issue = G_ < total_kt;
        const char *src = base + (long)kt * step;
        char *dst = smem + dbuf * BUF_BYTES + lds;
        auto piece = [&](auto pc) __attribute__((always_inline)) {
            constexpr int P = decltype(pc)::value;
            fence_sched();
            if (issue && P < np) // (wave-uniform)
                __builtin_amdgcn_global_load_lds(IROCM_GLB_PTR(src + (unsigned long)off[P]), IROCM_LDS_PTR(dst + P * 1024), 16, 0, 0);
            fence_sched();
        };
        if constexpr (qa == 0) {
            compute_ks(qac, I0{}, I2{}, I0{}, zeroc, aq, bq0);
            piece(I0{});
            compute_ks(qac, I0{}, I2{}, I1{}, zeroc, aq, bq0);
            piece(I1{});
            if constexpr (NJ1 > 0) compute_ks(qac, I1{}, IJ1{}, I0{}, zeroc, aq, bq1);
            piece(I2{});
            if constexpr (NJ1 > 0) compute_ks(qac, I1{}, IJ1{}, I1{}, zeroc, aq, bq1);
            piece(std::integral_constant<int, 3>{});
        } else {
            if constexpr (NJ1 > 0) compute_ks(qac, I1{}, IJ1{}, I0{}, zeroc, aq, bq1);
            piece(I0{});
            if constexpr (NJ1 > 0) compute_ks(qac, I1{}, IJ1{}, I1{}, zeroc, aq, bq1);
            piece(I1{});
            compute_ks(qac, I0{}, I2{}, I0{}, zeroc, aq, bq0);
            piece(I2{});
            compute_ks(qac, I0{}, I2{}, I1{}, zeroc, aq, bq0);
            piece(std::integral_constant<int, 3>{});
        }
        if (issue) { // advance; a cursor that crosses into the next tile finishes here, behind the MFMAs still in flight
            ++G_;
            if (++kt == nk) {
                kt = 0;
                if (++st < my_tiles)
                    cursor_cross(is_a, st, base, off);
            }
        }
        return issue;
    };
    auto ktile = [&](int buf, auto zeroc) {
        if constexpr (SCHED == 9) {
            const bool row0 = wr == 0;
            // L1
            stamp();
            read_b(I0{}, I2{}, bq0);
            if constexpr (NJ1 > 0) read_b(I1{}, IJ1{}, bq1);
            read_a(I0{}, aq);
            fine_stamp();
            wait_lgkm0();
            fine_stamp();
            barrier();
            // C1
            if constexpr (TRACE) { if (pa.trace_fine == 1 || pa.trace_fine == 4) stamp(); }
            __builtin_amdgcn_s_setprio(1);
            const bool c1_issue = burst9(I0{}, zeroc, std::true_type{}, row0 ? (buf ^ 1) : buf); // X: row 0 A(t + 1) -> the other buffer, row 1 B(t + 2) -> this one
            __builtin_amdgcn_s_setprio(0);
            fence_sched();
            if constexpr (TRACE) { if (pa.trace_fine == 1 || pa.trace_fine == 4) stamp(); }
            barrier();
            // L2
            stamp();
            read_a(I1{}, aq);
            flip_buf(buf ? -BUF_BYTES : BUF_BYTES);
            fine_stamp();
            if (!row0) { // row 1: everything older than the B pieces of its C1 has landed (its A one K-tile ahead included)
                if (c1_issue) wait_vm<NB>();
                else wait_vm<0>();
            }
            wait_lgkm0();
            fine_stamp();
            barrier();
            // C2
            if constexpr (TRACE) { if (pa.trace_fine == 1 || pa.trace_fine == 4) stamp(); }
            __builtin_amdgcn_s_setprio(1);
            const bool c2_issue = burst9(I1{}, zeroc, std::false_type{}, buf); // Y: row 0 B(t + 2), row 1 A(t + 2), both -> this buffer
            __builtin_amdgcn_s_setprio(0);
            fence_sched();
            if (row0) { // row 0: A(t + 1) (issued in C1) and everything older has landed; the B pieces just issued may still fly
                if (c2_issue) wait_vm<NB>();
                else wait_vm<0>();
            }
            if constexpr (TRACE) { if (pa.trace_fine == 1 || pa.trace_fine == 4) stamp(); }
            barrier();
            return;
        }
        // L1
        stamp();
        if constexpr (SCHED == 1) {
            if (a_G < total_kt)
                stage_a_next(buf ^ 1, std::false_type{}, std::true_type{});
        }
        if constexpr (SCHED == 3) {
            // 8 B reads | A pieces 0, 1 | 8 B reads | A pieces 2, 3 | 8 A reads
            const bool more = a_G < total_kt;
            char *dst = smem + (buf ^ 1) * BUF_BYTES;
            const char *src = a_base + (long)a_kt * a_step;
            read_b(I0{}, I2{}, bq0);
            if (more) stage_part<4, 0, 2>(src, a_off, dst, w);
            if constexpr (NJ1 > 0) read_b(I1{}, IJ1{}, bq1);
            if (more) stage_part<4, 2, 4>(src, a_off, dst, w);
            read_a(I0{}, aq);
            if (more) stage_a_next(buf ^ 1, std::false_type{}, std::false_type{}); // (cursor only)
        } else {
            read_b(I0{}, I2{}, bq0);
            if constexpr (NJ1 > 0) read_b(I1{}, IJ1{}, bq1);
            read_a(I0{}, aq);
        }
        if constexpr (SCHED == 0) {
            if (a_G < total_kt)
                stage_a_next(buf ^ 1, std::false_type{}, std::true_type{});
        }
        fine_stamp();
        wait_lgkm0();
        if constexpr (SCHED == 2) {
            if (a_G < total_kt) {
                stage_a_next(buf ^ 1, std::false_type{}, std::true_type{});
                wait_lgkm0(); // (a cursor that crossed into the next tile asked for its table entry: finish_cursors reads it in C1)
            }
        }
        fine_stamp();
        barrier();
        // C1
        if constexpr (TRACE) { if (pa.trace_fine == 1 || pa.trace_fine == 4) stamp(); }
        if constexpr (KV != 7 && KV != 8) __builtin_amdgcn_s_setprio(1);
        mask_b(I0{}, I2{}, bq0); // (tap mode; hipcc spreads the ANDs of the later fragments between the first MFMAs)
        if constexpr (NJ1 > 0) mask_b(I1{}, IJ1{}, bq1);
        compute(I0{}, I0{}, I2{}, zeroc, aq, bq0);
        if constexpr (NJ1 > 0) compute(I0{}, I1{}, IJ1{}, zeroc, aq, bq1);
        if constexpr (TAPS && !B_KMAJOR) {
            // order of this interval: the two masks the first eight MFMAs need (1 v_bfe + 4 v_and per column tile), then two mask
            // instructions in the shadow of each MFMA until all 4 NT + 1 per... are out (left alone hipcc issues every AND first:
            // ~30 VALU = 150-250 cycles in front of the burst, every K-tile)
            __builtin_amdgcn_sched_group_barrier(0x2, 10, 0);
            constexpr int kRest = NT * 9 - 10; // VALU left: NT bfe + 8 NT and, minus the ten in front
#pragma unroll
            for (int g = 0; g < (kRest + 1) / 2; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x8, NT * 8 - (kRest + 1) / 2, 0);
        }
        if constexpr (KV != 7 && KV != 8) __builtin_amdgcn_s_setprio(0);
        fence_sched();
        if constexpr (TRACE) { if (pa.trace_fine == 1 || pa.trace_fine == 4) stamp(); }
        finish_cursors(); // (issues behind the MFMA burst, which is still executing)
        barrier();
        // L2
        stamp();
        const bool b_more = b_G < total_kt; // (a B tile is issued in this phase)
        if constexpr (SCHED == 1) {
            if (b_more)
                stage_b_next(buf, std::false_type{}, std::true_type{});
        }
        if constexpr (SCHED == 3) {
            // 4 A reads | B pieces 0, 1 | 4 A reads | B pieces 2, 3  (read_a issues i-major: the first four reads are rows 0, 1)
            const bool more = b_more;
            char *dst = smem + buf * BUF_BYTES + OPER_BYTES;
            const char *src = b_base + (long)b_kt * b_step;
            if (more) stage_part<NB, 0, 2>(src, b_off, dst, w);
            read_a(I1{}, aq);
            if (more) stage_part<NB, 2, 4>(src, b_off, dst, w);
            if (more) stage_b_next(buf, std::false_type{}, std::false_type{}); // (cursor only)
        } else {
            read_a(I1{}, aq);
        }
        flip_buf(buf ? -BUF_BYTES : BUF_BYTES); // every read of this K-tile is issued
        if constexpr (SCHED == 0) {
            if (b_more)
                stage_b_next(buf, std::false_type{}, std::true_type{});
        }
        fine_stamp();
        if constexpr (SCHED == 2) {
            wait_lgkm0();
            if (b_more) {
                stage_b_next(buf, std::false_type{}, std::true_type{});
                wait_vm<NB>();
                wait_lgkm0();
            } else {
                wait_vm<0>();
            }
        } else {
            if (b_more) wait_vm<NBW>(); // everything older than these NB loads has landed (C stores of a previous tile included)
            else wait_vm<0>();
            wait_lgkm0();
        }
        fine_stamp();
        barrier();
        // C2
        if constexpr (TRACE) { if (pa.trace_fine == 1 || pa.trace_fine == 4) stamp(); }
        if constexpr (KV != 7 && KV != 8) __builtin_amdgcn_s_setprio(1);
        if constexpr (NJ1 > 0) compute(I1{}, I1{}, IJ1{}, zeroc, aq, bq1);
        compute(I1{}, I0{}, I2{}, zeroc, aq, bq0);
        if constexpr (KV != 7 && KV != 8) __builtin_amdgcn_s_setprio(0);
        if constexpr (TRACE) { if (pa.trace_fine == 1 || pa.trace_fine == 4) stamp(); }
        if constexpr (TAPS)
            c_t = c_t == 8 ? 0 : c_t + 1;
        barrier();
    };

    // (the first K-tiles were requested at the top of the kernel, before the address arithmetic and the accumulator clear)
    if constexpr ((CONV == 0) && IROCM_KV == 9) {
        if (total_kt > 1) {
            if (wr == 1) wait_vm<NB + 4>(); // (B(1) and its own A(1) may still fly)
            else wait_vm<NB>();
        } else {
            wait_vm<0>();
        }
    } else {
        if (total_kt > 1)
            wait_vm<NBW>();
        else
            wait_vm<0>();
    }
    wait_lgkm0(); // (the tile table's ds_write)
    barrier();
    if constexpr ((CONV == 0) && IROCM_KV == 9) { // schedule 9: the generic cursors take over (by value)
        const bool r0 = wr == 0;
        x_is_a = r0;
        x_base = r0 ? a_base : b_base;
        y_base = r0 ? b_base : a_base;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x_off[i] = r0 ? a_off[i] : b_off[i];
            y_off[i] = r0 ? b_off[i] : a_off[i];
        }
        x_kt = r0 ? a_kt : b_kt; y_kt = r0 ? b_kt : a_kt;
        x_s = r0 ? a_s : b_s;    y_s = r0 ? b_s : a_s;
        x_G = r0 ? a_G : b_G;    y_G = r0 ? b_G : a_G;
        x_step = r0 ? a_step : b_step; y_step = r0 ? b_step : a_step;
        x_np = r0 ? 4 : NB;      y_np = r0 ? NB : 4;
        x_lds = r0 ? w * 4 * 1024 : OPER_BYTES + w * NB * 1024;
        y_lds = r0 ? OPER_BYTES + w * NB * 1024 : w * 4 * 1024;
        // (the prologue's stage calls decode in place: no table request is pending here)
    }
    if (wr == 1)
        barrier(); // stagger: wave row 1 runs one barrier interval behind wave row 0
    if constexpr (KV == 7) { // static priority for the younger half (MI355X_MICROARCH.md, two waves per SIMD, item 4), no per-phase flips
        if (wr == 1) __builtin_amdgcn_s_setprio(1);
    }
    // Two nested loops over the SAME flat pipeline: the inner K loop stays a compact body with a short back edge (an
    // epilogue inlined into one flat loop pushes the back edge beyond the +-128 KB reach of s_cbranch: every K-tile then
    // pays s_getpc / s_setpc trampolines through cold code — measured 3-4 % slower than the one-shot kernel).
    int G = 0;
    for (int c_s = 0; c_s < my_tiles; ++c_s) {
        int c_ib, c_m0, c_n0; // the tile being accumulated
        if (c_s < tab_n) {
            unsigned e0 = tab_read(c_s, 0), e1 = tab_read(c_s, 1);
            wait_lgkm0();
            asm volatile("" : "+v"(e0), "+v"(e1));
            e0 = (unsigned)__builtin_amdgcn_readfirstlane((int)e0);
            e1 = (unsigned)__builtin_amdgcn_readfirstlane((int)e1);
            c_ib = (int)(e0 >> 16);
            c_m0 = (int)(e0 & 0xffffu) * BM;
            c_n0 = (int)(e1 & 0xffffu) * BN_;
        } else {
            decode(c_s, c_ib, c_m0, c_n0);
        }
        // (split-K of the tap mode: the accumulators' row blocks are rotated by rot — block 0 is filter block rot — and the slice stores
        // blocks 0 .. 8 / S - 1 only, so bias rows and store rows simply start rot * 16 rows further down)
        const int c_rot16 = (CONV == 3 && S > 1) ? sp_slice * (8 / S) * 16 : 0;
        if constexpr (CONV != 0)
            load_cbias(c_m0 + c_rot16);
        if constexpr (TAPS) {
            // validity bits of this tile's slots (nine per column tile and lane; rebuilt from an opaque lane id: see epilogue)
            int lane_o = (int)(threadIdx.x & 63);
            asm volatile("" : "+v"(lane_o));
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const unsigned slot = (unsigned)(c_n0 + wc * (16 * NT) + j * 16 + (lane_o & 15));
                unsigned img, pix, oy, ox;
                udivmod_m(slot, (unsigned)p.cv_hwp, p.cv_hwp_m, img, pix);
                udivmod_m(pix, (unsigned)p.cv_ow, p.cv_ow_m, oy, ox);
                const unsigned rowm = ((int)oy >= p.cv_ylo ? 1u : 0u) | 2u | ((int)oy < p.cv_yhi ? 4u : 0u);
                const unsigned colm = ((int)ox >= p.cv_xlo ? 1u : 0u) | 2u | ((int)ox < p.cv_xhi ? 4u : 0u);
                tv[j] = ((rowm & 1u) ? colm : 0u) | (colm << 3) | ((rowm & 4u) ? (colm << 6) : 0u);
            }
            c_t = kt0 % 9;
        }
        if constexpr (CONV == 0) {
            if (p.bias != nullptr && c_n0 + BN_ <= p.n) { // (wave-uniform)
                const char *bb = (const char *)((const unsigned short *)p.bias + (long)c_ib * p.bias_b + c_n0);
                if (((((uintptr_t)bb) & 15) == 0) && w == 0) {
                    int lo = lane; // (opaque: see the epilogue's bias read)
                    asm volatile("" : "+v"(lo));
                    __builtin_amdgcn_global_load_lds(IROCM_GLB_PTR(bb + (unsigned long)(unsigned)min(lo * 16, BN_ * 2 - 16)),
                                                     IROCM_LDS_PTR(smem + LDS_BYTES + (TRACE ? kTraceBytes : 0) + kTabBytes + (c_s & 1) * kBiasSlotBytes),
                                                     16, 0, 0);
                }
            }
        }
        ktile(G & 1, std::true_type{}); // (the tile's first K-tile starts the accumulators from zero)
        ++G;
        for (int kt = 1; kt < nk; ++kt, ++G)
            ktile(G & 1, std::false_type{});
        stamp();
        // Wave row 0 reaches here one barrier interval before row 1. Without the two barriers below the rows' epilogues
        // serialise: row 1's last (short) compute interval lasts as long as row 0's epilogue, and row 0's first compute of
        // the next tile as long as row 1's — 2 x 9.4 k cycles per tile boundary in the timeline (tools/gemm_timeline.py).
        // Row 0 waits out row 1's last compute, both rows run their epilogues side by side (10-11 k cycles for both), row 1
        // falls back one interval.
        if (wr == 0)
            barrier();
        stamp(); // (TRACE: row 0's wait for row 1)
        // this wave's part of tile c_s is complete; G = first K-tile of the next tile
        if constexpr (CONV != 0) { // the conv launcher admits act 0 / 1 only
            using R = std::integral_constant<bool, CONV == 2>; // (tap mode: no residual copy)
            unsigned rbmask = 0xffu;
            if constexpr (CONV == 3) {
                if (S > 1)
                    rbmask = exchange();
            }
            if (p.act == 0) epilogue_conv(std::integral_constant<int, 0>{}, R{}, c_m0 + c_rot16, c_n0, rbmask);
            else epilogue_conv(std::integral_constant<int, 1>{}, R{}, c_m0 + c_rot16, c_n0, rbmask);
        } else {
            if (p.act == 0) epilogue(std::integral_constant<int, 0>{}, c_ib, c_m0, c_n0, c_s & 1);
            else if (p.act == 1) epilogue(std::integral_constant<int, 1>{}, c_ib, c_m0, c_n0, c_s & 1);
            else epilogue(std::integral_constant<int, 5>{}, c_ib, c_m0, c_n0, c_s & 1); // launch_p admits act 0, 1, 5 only
        }
        if (wr == 1)
            barrier();
        stamp();
    }
    if (wr == 0)
        barrier(); // balance the stagger
    if constexpr (TRACE) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        stamp();
        __syncthreads();
        for (int i = lane; i < kTraceSlots; i += 64)
            pa.trace[((size_t)blockIdx.x * 8 + w) * kTraceSlots + i] =
                i < tslot ? *(unsigned long long *)(smem + LDS_BYTES + w * (kTraceSlots * 8) + i * 8) : 0ull;
    }
}

int launch_gemm256p_nt4(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, bool akm, bool bkm);
int launch_gemm256p_trace(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, int nt, unsigned long long *trace);
int launch_gemm256p_nt3(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, bool akm, bool bkm);
int launch_gemm256p_nt2(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, bool akm, bool bkm);

template <typename Tr, int NT, bool TRACE = false>
static int launch_p(infiniRocmRuntime_t rt, GemmArgs g, bool akm, bool bkm, unsigned long long *trace = nullptr) {
    PArgs pa;
    pa.trace = trace;
    pa.trace_fine = (TRACE && getenv("IROCM_GEMM_TRACE_FINE")) ? atoi(getenv("IROCM_GEMM_TRACE_FINE")) : 0;
    pa.split = 1; pa.slab = nullptr; pa.slab_bytes = 0; pa.flags = nullptr; pa.err = nullptr;
    constexpr int kLds = LDS_BYTES + (TRACE ? kTraceBytes : 0) + kExtraLds;
    if (!(g.act == 0 || g.act == 1 || g.act == 5) || (g.bias && !(g.bias_m == 0 && g.bias_n == 1)))
        IROCM_FAIL(INFINI_ROCM_UNSUPPORTED, "gemm256p: activation %d / this bias layout is not served by the persistent kernels", g.act);
    g.tiles_m = (int)ceil_div(g.m, BM);
    g.tiles_n = (int)ceil_div(g.n, 64 * NT);
    const long total = (long)g.tiles_m * g.tiles_n * g.batch;
    if (total >= (1l << 31))
        IROCM_FAIL(INFINI_ROCM_INVALID_ARGUMENT, "matmul: too many tiles");
    pa.g = g;
    pa.total_tiles = (int)total;
    pa.tab_n = (g.tiles_m < 65536 && g.tiles_n < 65536 && g.batch < 65536) ? kTileTab : 0;
    if (const char *cap = getenv("IROCM_GEMM_TAB_N")) { // test hook (read per call): a shorter table exercises the in-place decode of
        const int v = atoi(cap);                        // the steps beyond it, which production shapes reach only past 65 536 tiles
        if (v >= 0 && v < pa.tab_n)
            pa.tab_n = v;
    }
    pa.per_batch_m = udiv_magic((unsigned long long)g.tiles_m * g.tiles_n);
    pa.per_group_m = udiv_magic(8ull * g.tiles_n);
    // one workgroup per CU walking its tiles; gridDim.x % 8 == 0 keeps every workgroup's tiles on its XCD's id range
    unsigned grid = (unsigned)total;
    const unsigned cus = (unsigned)(rt->num_cu >= 8 ? (rt->num_cu / 8) * 8 : rt->num_cu);
    if (grid > cus)
        grid = cus;
#define IROCM_G256P(AK, BK_)                                                                       \
    do {                                                                                           \
        auto kern = gemm256p_kernel<Tr, AK, BK_, NT, TRACE>;                                       \
        IROCM_LDS_ATTR(kern, kLds, rt);                                                            \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), kLds, rt->stream, pa);                     \
    } while (0)
    if constexpr (TRACE) { // the timeline build exists for the ONNX "NN" layout only
        if (!(akm && !bkm))
            IROCM_FAIL(INFINI_ROCM_UNSUPPORTED, "gemm timeline: NN layout only");
        IROCM_G256P(true, false);
    } else {
        if (akm && bkm) IROCM_G256P(true, true);
        else if (akm && !bkm) IROCM_G256P(true, false);
        else if (!akm && bkm) IROCM_G256P(false, true);
        else IROCM_G256P(false, false);
    }
#undef IROCM_G256P
    IROCM_LAUNCH_CHECK("gemm256p");
    return INFINI_ROCM_OK;
}

// conv mode: one instantiation per tile width and residual flag (A K-major, B gathered by pixel slots)
// split > 1 (tap mode only): `split` workgroups per tile trade partial sums through `slab` (>= conv_tap_slab_bytes) / `flags`.
template <typename Tr, int NT, bool RES, bool TAPS = false, bool TRACE = false>
static int launch_p_conv(infiniRocmRuntime_t rt, GemmArgs g, int split = 1, char *slab = nullptr, size_t slab_bytes = 0, unsigned *flags = nullptr,
                         unsigned long long *trace = nullptr) {
    static_assert(!(RES && TAPS), "tap mode has no residual copy");
    PArgs pa;
    pa.trace = trace;
    pa.trace_fine = 0;
    pa.split = TAPS ? split : 1; pa.slab = slab; pa.slab_bytes = (unsigned)slab_bytes; pa.flags = flags; pa.err = rt->sync_err_dev;
    g.tiles_m = (int)ceil_div(g.m, BM);
    g.tiles_n = (int)ceil_div(g.n, 64 * NT);
    const long total = (long)g.tiles_m * g.tiles_n;
    if (total >= (1l << 31))
        IROCM_FAIL(INFINI_ROCM_INVALID_ARGUMENT, "conv: too many tiles");
    pa.g = g;
    pa.g.cv_hwp_m = udiv_magic((unsigned long long)g.cv_hwp);
    pa.total_tiles = (int)total;
    pa.tab_n = (g.tiles_m < 65536 && g.tiles_n < 65536) ? kTileTab : 0;
    pa.per_batch_m = udiv_magic((unsigned long long)g.tiles_m * g.tiles_n);
    pa.per_group_m = udiv_magic(8ull * g.tiles_n);
    unsigned grid = (unsigned)total;
    const unsigned cus = (unsigned)(rt->num_cu >= 8 ? (rt->num_cu / 8) * 8 : rt->num_cu);
    if (grid > cus)
        grid = cus;
    if (pa.split > 1) {
        // one unit per workgroup, the slices of a tile on one XCD: workgroup 8 idx + xcd = (tile (idx / S) * 8 + xcd, slice idx % S)
        grid = (unsigned)(ceil_div(total, 8) * 8 * pa.split);
        if (grid > cus || (g.k / BK) % pa.split != 0 || !(pa.split == 2 || pa.split == 4) || !slab || !flags)
            IROCM_FAIL(INFINI_ROCM_INVALID_ARGUMENT, "conv tap GEMM: split %d of %ld tiles is not launchable on %u CUs", pa.split, total, cus);
    }
    auto kern = gemm256p_kernel<Tr, true, false, NT, TRACE, TAPS ? 3 : (RES ? 2 : 1)>;
    constexpr int kLds = LDS_BYTES + (TRACE ? kTraceBytes : 0) + kExtraLds;
    IROCM_LDS_ATTR(kern, kLds, rt);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), kLds, rt->stream, pa);
    IROCM_LAUNCH_CHECK("gemm256p(conv)");
    return INFINI_ROCM_OK;
}

} // namespace g256p
} // namespace irocm
