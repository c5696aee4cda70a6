// Fused prefill attention for gfx950:  O = softmax(scale * Q K^T + mask) V   per (batch x head), f16 / bf16, head dim 64 / 128.
//
// Replaces the five-operator chain the ONNX front-end leaves in a transformer graph — MatMul(Q, K^T) -> Div/Mul(scalar) ->
// Add(mask) -> Softmax -> MatMul(P, V) (reference kernels: matmul.cc:67-174, element_wise.cu, softmax.cu; BASELINE config 4
// "Attention MatMul+Softmax fused path") — without ever writing the [Sq, Sk] score matrix to HBM
// (BERT-base bs32 seq512: 5 x 201 MB of traffic per layer -> Q, K, V, O only).
//
// One workgroup = 4 waves = 4 * QW query rows of one (batch, head); keys are swept in tiles of 64 with the running
// max / sum recurrence of online softmax (fp32, base-2 exponent with scale * log2(e) folded in).
//   * Both contractions run on v_mfma_f32_16x16x32 with SWAPPED roles so that every lane owns one query column:
//       S^T[key, q] = K Q^T   (A = K rows from LDS via ds_read_b128, B = Q fragments kept in registers for the whole sweep)
//       O^T[d, q]  += V^T P^T (A = V^T via ds_read_b64_tr_b16 from the [key][d] LDS image, B = P packed from the S^T
//                              accumulators IN REGISTERS: a lane's 4 + 4 scores of two 16-key tiles are exactly the 8
//                              k-slots of its B fragment once the V rows are fetched in the same permuted key order)
//     so the row statistics, the rescale of O and the final 1/l are per-lane scalars and P never touches LDS.
//   * K / V tiles: register-staged double buffering (global loads of tile t+1 fly during the MFMAs of tile t), LDS images
//     padded to conflict-free pitches (K: D*2 + 16 B for ds_read_b128, V: D*2 + 32 B for the transpose reads).
//   * mask: additive, one value per (batch-group, key) broadcast over the query rows (the BERT [B,1,1,S] padding mask) —
//     its 64 values per key tile ride with the K / V tile into LDS as fp32 (/ scale, -inf past Sk) and enter the score
//     MFMAs as their C operand; the scale itself is applied inside the exponent's fma;
//     optional causal masking (bottom-right aligned: key > query + Sk - Sq -> -inf). Keys past Sk are -inf. Fully masked rows produce 0.
//   * scale comes from device memory (the graph's scalar constant), multiply or divide — the launch stays capturable.
#include "gemm_common.h"

namespace irocm {

struct AttnArgs {
    const void *q, *k, *v, *mask, *scale;
    void *o;
    int bh, sq, sk;
    int mask_group;   // mask row index = bh / mask_group (heads per mask row); mask row stride = sk
    int scale_is_div; // effective scale = scale_value (0) or 1 / scale_value (1)
    float scale_imm;  // used when scale == nullptr
    int causal;
    int mask_rows;    // 0: mask is per key [G][sk]; sq: a full additive mask [G][sq][sk] (e.g. a causal mask passed as a tensor)
    int o_heads;      // > 0: O is stored merged, [bh / o_heads][sq][o_heads][D] (Transpose(0,2,1,3) -> Reshape of the plain result)
};

// raw v_exp_f32 (exp2f() wraps it in a denormal range fix-up: 5 extra VALU instructions per score); results below
// 2^-126 flush to 0, which is what a softmax weight that small is worth
__device__ static inline float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Cross-lane maximum / sum over the four 16-lane groups of a wave (a query column's keys are spread over lanes l15 + 16 g4) WITHOUT the
// LDS crossbar: v_permlane16_swap / v_permlane32_swap hand every lane its partner's value as one VALU operation each. __shfl_xor(x, 16 / 32)
// compiles to ds_bpermute_b32 + s_waitcnt lgkmcnt(0) — two serialised LDS round trips (and a drain of every other LDS request in flight)
// on the dependency chain between the score MFMAs and the exponentials of EVERY key tile and query tile (round 6).
// Returns {x of this lane, x of lane ^ 16} resp. ^ 32 in some order, as two values the caller combines (max / add).
// Inline asm, not __builtin_amdgcn_permlane16_swap: handed two copies of ONE value the ROCm 7.2 compiler folds the builtin's second
// result into its first (x + x, max(x, x): a "reduction" that reduces nothing — caught by the causal D = 128 parity case, where the
// row sum went through it). The s_nop covers the VALU-write -> permlane-swap-read hazard the compiler cannot see inside the asm.
__device__ static inline void lanes_xor16(float x, float &a, float &b) {
    a = x;
    b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ static inline void lanes_xor32(float x, float &a, float &b) {
    a = x;
    b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// CAUSAL / MASK are compile-time so that the unmasked, non-causal sweep carries no select / compare per score.
template <typename Tr, int D, int NT, bool CAUSAL, int MASK> // MASK: 0 none, 1 per key, 2 per (query, key)
__global__ __launch_bounds__(256, (NT == 2 && D == 64) ? 3 : 2) void attention_kernel(AttnArgs p) {
    constexpr int QW = NT * 16;              // query rows per wave
    constexpr int KT = 64;                   // keys per tile
    // K image pitch (bytes): + 32, not + 16 — ds_read_b128 is serviced in the lane groups {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH,
    // LDS): with a 36- / 68-dword pitch lanes l15 = 4 (g4 = 1) and l15 = 13 (g4 = 0) of one group share banks (2-way conflicts on
    // every K-fragment read: 20 % of the kernel's LDS cycles in profiles/r03_attention_bert_pmc.json); 40 / 72 dwords are conflict-free
    constexpr int KP = D * 2 + 32;
    constexpr int VP = D * 2 + 32;           // V image pitch (bytes)
    constexpr int K_BYTES = KT * KP, V_BYTES = KT * VP, STAGE = K_BYTES + V_BYTES;
    constexpr int NCH = KT * (D / 8) / 256;  // 16-byte runs per thread per operand tile
    constexpr int KS = D / 32;               // k-steps of the S contraction
    constexpr int DT = D / 16;               // d tiles of O^T
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int l15 = lane & 15, g4 = lane >> 4;
    // XCD-aware placement: workgroups are dealt round-robin to the 8 XCDs in dispatch order (x fastest), so the query blocks
    // of one (batch, head) would land on different XCDs and each XCD's L2 would fetch that head's K / V from HBM again —
    // measured on the BERT shape (4 query blocks per head): 252 MB moved for 101 MB of Q, K, V, O, i.e. 5 TB/s: the kernel
    // was HBM-bound on its own re-reads. xcd_remap hands every XCD a contiguous range of (head, query block) pairs.
    const unsigned vid = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int bh = (int)(vid / gridDim.x), qblk = (int)(vid % gridDim.x);
    const int q0 = qblk * (4 * QW) + w * QW;
    const unsigned short *Q = (const unsigned short *)p.q + (long)bh * p.sq * D;
    const unsigned short *K = (const unsigned short *)p.k + (long)bh * p.sk * D;
    const unsigned short *V = (const unsigned short *)p.v + (long)bh * p.sk * D;
    const unsigned short *M = p.mask ? (const unsigned short *)p.mask + (long)(bh / p.mask_group) * p.sk * (MASK == 2 ? p.sq : 1) : nullptr;

    float scale = p.scale_imm;
    if (p.scale) {
        const float sv = Tr::to_f32(*(const unsigned short *)p.scale);
        scale = p.scale_is_div ? 1.0f / sv : sv;
    }
    constexpr float LOG2E = 1.4426950408889634f;
    // The scores stay in raw q.k units ("S' units") until the exponent: S' = K Q^T + mask / scale is what the MFMA chain
    // produces (the additive term is its C operand), and P = exp2(S' c - m' c) is one fma per score. That needs c > 0 for
    // the running maximum to commute with the scale: the sign of a negative scale goes into Q (exact), a zero scale zeroes Q.
    const bool neg_q = scale < 0.f, zero_q = scale == 0.f;
    const float sabs = zero_q ? 1.0f : fabsf(scale);
    const float c = sabs * LOG2E;
    const float inv_scale = 1.0f / sabs; // additive mask (natural-log units) -> S' units
    const bool mask_vec = MASK != 0 && (p.sk % 4 == 0) && ((((uintptr_t)p.mask) & 7) == 0);

    // Q fragments (B operand: lane = query column l15, 8 consecutive d)
    s16x8_t qf[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        int qr = q0 + nt * 16 + l15;
        qr = qr < p.sq ? qr : p.sq - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[nt][ks] = *(const s16x8_t *)(Q + (long)qr * D + ks * 32 + g4 * 8);
    }
    if (neg_q || zero_q) { // wave-uniform, never in a transformer graph (scale = 1 / sqrt(D))
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    qf[nt][ks][e] = zero_q ? (short)0 : (short)(qf[nt][ks][e] ^ (short)0x8000);
    }

    // staging assignment
    int st_row[NCH], st_col[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = t + i * 256;
        st_row[i] = ch / (D / 8);
        st_col[i] = (ch % (D / 8)) * 8;
    }
    s16x8_t kreg[NCH], vreg[NCH];
    // The additive term of a key tile — MASK == 1: mask[key] / scale, otherwise 0; -inf for keys past Sk — travels with the
    // K / V tile: wave 0 fetches the mask values with the next tile's operands and leaves the 64 terms in LDS as fp32, and a
    // tile reads its 16 per lane with four ds_read_b128 as the C operands of its first MFMAs: no global load, convert,
    // compare or accumulator clear inside the tile. (A mask load issued inside the tile is waited for with vmcnt(0) right
    // after the S product — hipcc cannot count across the branches of a guarded load — and that wait also drains the K / V
    // prefetch: the masked BERT shape ran 61 us against 53 us unmasked.)
    unsigned short mreg = 0;
    auto load_tile = [&](int kbase) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int kr = kbase + st_row[i];
            kr = kr < p.sk ? kr : p.sk - 1; // clamped rows are masked to -inf below
            kreg[i] = *(const s16x8_t *)(K + (long)kr * D + st_col[i]);
            vreg[i] = *(const s16x8_t *)(V + (long)kr * D + st_col[i]);
        }
        if constexpr (MASK == 1) {
            if (t < KT)
                mreg = M[kbase + t < p.sk ? kbase + t : p.sk - 1];
        }
    };
    auto store_tile = [&](char *stage, float *mstrip, int kbase) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            *(s16x8_t *)(stage + st_row[i] * KP + st_col[i] * 2) = kreg[i];
            *(s16x8_t *)(stage + K_BYTES + st_row[i] * VP + st_col[i] * 2) = vreg[i];
        }
        if (t < KT)
            mstrip[t] = kbase + t < p.sk ? (MASK == 1 ? Tr::to_f32(mreg) * inv_scale : 0.f) : -INFINITY;
    };
    float *const mstrips = (float *)(smem + 2 * STAGE); // 2 x KT floats

    f32x4 o[DT][NT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            o[dt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[NT], l_run[NT]; // running max (S' units, reduced over the 4 lanes of a column) and per-lane partial sum
    // MMA_SUM (D = 64, round 4): the row sum l = sum_k P[k][q] is a 17th output row of the P V product — one more MFMA tile per
    // (query tile, key half) against an all-ones A operand — instead of 16 v_add per query tile and key tile on the VALU pipe, which
    // is the busy one (59 % against 21 % for the matrix pipe, profiles/r03_attention_bert_pmc.json). The sum is then taken over
    // the ROUNDED weights (the values the P V product really uses) and needs no cross-lane reduction at the end. D = 128 has no
    // registers to spare (240-256 VGPRs) and keeps the VALU sum.
    constexpr bool MMA_SUM = D == 64 && !(CAUSAL && MASK == 2); // (that variant sits at the 168-register limit already: it would spill)
    f32x4 osum[MMA_SUM ? NT : 1];
    s16x8_t ones;
#pragma unroll
    for (int e = 0; e < 8; ++e)
        ones[e] = (short)Tr::from_f32(1.0f);
    if constexpr (MMA_SUM) { // opaque to the compiler: otherwise it re-materialises the constant with four v_mov in front of every MFMA
        u32x4_t ow = __builtin_bit_cast(u32x4_t, ones);
        asm volatile("" : "+v"(ow));
        ones = __builtin_bit_cast(s16x8_t, ow);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        m_run[nt] = -INFINITY;
        l_run[nt] = 0.f;
        if constexpr (MMA_SUM)
            osum[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // key tiles this workgroup needs (causal: nothing past its last query row)
    int nkt = (p.sk + KT - 1) / KT;
    if (p.causal) {
        const int last_q = min(p.sq, (qblk + 1) * 4 * QW) - 1;
        nkt = max(1, min(nkt, (last_q + p.sk - p.sq) / KT + 1));
    }

    load_tile(0);
    store_tile(smem, mstrips, 0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const char *cur = smem + (kt & 1) * STAGE;
        const int kbase = kt * KT;
        if (kt + 1 < nkt)
            load_tile(kbase + KT);

        // ---- S'^T = K Q^T + term : 4 key sub-tiles x NT query tiles (lane: query column l15, keys kbase + mt*16 + 4*g4 + r)
        const float *mstrip = mstrips + (kt & 1) * KT;
        f32x4 s[4][NT];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            s16x8_t kf[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                kf[ks] = *(const s16x8_t *)(cur + (mt * 16 + l15) * KP + (ks * 32 + g4 * 8) * 2);
            const f32x4 term = *(const f32x4 *)(mstrip + mt * 16 + 4 * g4);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 a = term;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    a = Tr::mfma(kf[ks], qf[nt][ks], a);
                s[mt][nt] = a;
            }
        }
        if constexpr (MASK == 2) {
            // full additive mask: the lane's query column has its own mask row; 4 consecutive keys = one 8-byte load
            const bool ragged = kbase + KT > p.sk; // wave-uniform
            const bool mvec = !ragged && mask_vec;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                int qr = q0 + nt * 16 + l15;
                qr = qr < p.sq ? qr : p.sq - 1; // rows past Sq are computed on the last row and never stored
                const unsigned short *Mr = M + (long)qr * p.sk;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int key = kbase + mt * 16 + 4 * g4;
                    unsigned short mh[4] = {0, 0, 0, 0};
                    if (mvec) {
                        const u32x2_t mk = *(const u32x2_t *)(Mr + key);
                        mh[0] = (unsigned short)(mk[0] & 0xffff); mh[1] = (unsigned short)(mk[0] >> 16);
                        mh[2] = (unsigned short)(mk[1] & 0xffff); mh[3] = (unsigned short)(mk[1] >> 16);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (key + r < p.sk)
                                mh[r] = Mr[key + r];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        s[mt][nt][r] = fmaf(Tr::to_f32(mh[r]), inv_scale, s[mt][nt][r]); // keys past Sk are -inf already
                }
            }
        }
        if (CAUSAL) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int lim = q0 + nt * 16 + l15 + (p.sk - p.sq); // last admissible key of this query column
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kbase + mt * 16 + 4 * g4 + r > lim)
                            s[mt][nt][r] = -INFINITY;
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float mx = -INFINITY;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    mx = fmaxf(mx, s[mt][nt][r]);
            float m_new;
            // (the causal + masked D = 64 copies sit at the 168-register limit of three waves per SIMD: the swap's register pairs
            // spilled three registers there — they keep the LDS crossbar form; tests/test_kernel_resources_cpu.py watches this)
            if constexpr (CAUSAL && D == 64 && MASK != 0) {
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                m_new = fmaxf(m_run[nt], mx);
            } else {
                float xa, xb;
                lanes_xor16(mx, xa, xb);
                lanes_xor32(fmaxf(xa, xb), xa, xb);
                m_new = fmaxf(m_run[nt], fmaxf(xa, xb)); // (one v_max3_f32)
            }
            const float m_use = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = fast_exp2((m_run[nt] - m_use) * c);
            const float nmc = -m_use * c;
            m_run[nt] = m_new;
            float ps = 0.f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = fast_exp2(fmaf(s[mt][nt][r], c, nmc));
                    s[mt][nt][r] = e;
                    if constexpr (!MMA_SUM)
                        ps += e;
                }
            if constexpr (!MMA_SUM)
                l_run[nt] = l_run[nt] * alpha + ps;
            // the running maximum settles after the first tiles: skip the 16 multiplies when no lane's changed
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    o[dt][nt] *= alpha;
                if constexpr (MMA_SUM)
                    osum[nt] *= alpha;
            }
        }
        // ---- P fragments (B operand): k-slots 0..3 = keys 4*g4 + e of sub-tile 2*kk, 4..7 = of sub-tile 2*kk+1 ----
        s16x8_t pf[NT][2];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                s16x8_t f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f[e] = (short)Tr::from_f32(s[2 * kk][nt][e]);
                    f[4 + e] = (short)Tr::from_f32(s[2 * kk + 1][nt][e]);
                }
                pf[nt][kk] = f;
            }
        // ---- O^T += V^T P^T ---------------------------------------------------------------------------
        const char *vimg = cur + K_BYTES;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                // transpose read: lane p of a 16-lane group supplies key row (p >> 2) of its 4-row block and
                // 4 d-columns (p & 3) * 4, receives d-column p of that block
                const int d4 = dt * 16 + (l15 & 3) * 4;
                const char *a0 = vimg + (kk * 32 + 4 * g4 + (l15 >> 2)) * VP + d4 * 2;
                const char *a1 = a0 + 16 * VP;
                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t *)(a0));
                const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t *)(a1));
                const s16x8_t vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    o[dt][nt] = Tr::mfma(vf, pf[nt][kk], o[dt][nt]);
            }
        if constexpr (MMA_SUM) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    osum[nt] = Tr::mfma(ones, pf[nt][kk], osum[nt]);
        }
        if (kt + 1 < nkt)
            store_tile(smem + ((kt + 1) & 1) * STAGE, mstrips + ((kt + 1) & 1) * KT, kbase + KT);
        __syncthreads();
    }

    // ---- normalise and store: lane holds O[q = .. + l15][d = dt*16 + 4*g4 + r] ------------------------------
    // plain: O[bh][q][D]; merged (o_heads = H): O[bh / H][q][bh % H][D] — the head merge a transformer layer applies next
    const long o_row = p.o_heads ? (long)p.o_heads * D : (long)D;
    unsigned short *O = (unsigned short *)p.o +
                        (p.o_heads ? ((long)(bh / p.o_heads) * p.sq * p.o_heads + (bh % p.o_heads)) * D : (long)bh * p.sq * D);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        float l;
        if constexpr (MMA_SUM) {
            l = osum[nt][0]; // every row of the ones tile holds the column's sum over all 64 keys of every tile
        } else {
            float xa, xb;
            lanes_xor16(l_run[nt], xa, xb);
            lanes_xor32(xa + xb, xa, xb);
            l = xa + xb;
        }
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        const int qi = q0 + nt * 16 + l15;
        if (qi >= p.sq)
            continue;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            u32x2_t pk;
            pk[0] = (unsigned)Tr::from_f32(o[dt][nt][0] * inv) | ((unsigned)Tr::from_f32(o[dt][nt][1] * inv) << 16);
            pk[1] = (unsigned)Tr::from_f32(o[dt][nt][2] * inv) | ((unsigned)Tr::from_f32(o[dt][nt][3] * inv) << 16);
            *(u32x2_t *)(O + (long)qi * o_row + dt * 16 + 4 * g4) = pk;
        }
    }
}

template <typename Tr, int D, int NT, bool CAUSAL, int MASK>
static int launch_attn2(infiniRocmRuntime_t rt, const AttnArgs &p) {
    constexpr int LDS = 2 * (64 * (D * 2 + 32) + 64 * (D * 2 + 32)) + 2 * 64 * 4;
    auto kern = attention_kernel<Tr, D, NT, CAUSAL, MASK>;
    IROCM_LDS_ATTR(kern, LDS, rt);
    dim3 grid((unsigned)ceil_div(p.sq, 4 * NT * 16), (unsigned)p.bh);
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS, rt->stream, p);
    IROCM_LAUNCH_CHECK("attention");
    return INFINI_ROCM_OK;
}
template <typename Tr, int D, int NT> static int launch_attn(infiniRocmRuntime_t rt, const AttnArgs &p) {
    const int mk = p.mask ? (p.mask_rows ? 2 : 1) : 0;
    if (p.causal)
        return mk == 2 ? launch_attn2<Tr, D, NT, true, 2>(rt, p)
                       : (mk == 1 ? launch_attn2<Tr, D, NT, true, 1>(rt, p) : launch_attn2<Tr, D, NT, true, 0>(rt, p));
    return mk == 2 ? launch_attn2<Tr, D, NT, false, 2>(rt, p)
                   : (mk == 1 ? launch_attn2<Tr, D, NT, false, 1>(rt, p) : launch_attn2<Tr, D, NT, false, 0>(rt, p));
}

} // namespace irocm

using namespace irocm;

extern "C" int infini_rocm_attention(infiniRocmRuntime_t rt, int dtype, const void *q, const void *k, const void *v,
                                     const void *mask, void *out, int64_t batch_heads, int64_t seq_q, int64_t seq_k,
                                     int64_t head_dim, int64_t mask_group, const void *scale_dev, int scale_is_div,
                                     float scale, int causal) {
    return infini_rocm_attention_ex(rt, dtype, q, k, v, mask, out, batch_heads, seq_q, seq_k, head_dim, mask_group,
                                    scale_dev, scale_is_div, scale, causal, 0, 0);
}

extern "C" int infini_rocm_attention_ex(infiniRocmRuntime_t rt, int dtype, const void *q, const void *k, const void *v,
                                        const void *mask, void *out, int64_t batch_heads, int64_t seq_q, int64_t seq_k,
                                        int64_t head_dim, int64_t mask_group, const void *scale_dev, int scale_is_div,
                                        float scale, int causal, int64_t heads, int mask_2d) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(heads >= 0 && heads < 65536 && (heads == 0 || batch_heads % heads == 0),
                    "attention: heads %lld does not divide batch_heads %lld", (long long)heads, (long long)batch_heads);
    IROCM_CHECK_ARG(dtype == INFINI_DT_F16 || dtype == INFINI_DT_BF16, "attention: f16 / bf16 only (got %s)", dtype_name(dtype));
    IROCM_CHECK_ARG(head_dim == 64 || head_dim == 128, "attention: head dim %lld not in {64, 128}", (long long)head_dim);
    IROCM_CHECK_ARG(batch_heads >= 0 && seq_q >= 0 && seq_k > 0 && batch_heads < 65536 && seq_q < (1ll << 31) && seq_k < (1ll << 31),
                    "attention: bad extent");
    IROCM_CHECK_ARG(!mask || mask_group > 0, "attention: mask_group must be positive");
    if (batch_heads == 0 || seq_q == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(q && k && v && out, "attention: NULL tensor");
    IROCM_CHECK_ARG(((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v)) & 15) == 0 && (((uintptr_t)out) & 7) == 0,
                    "attention: operands must be 16-byte aligned");
    AttnArgs p;
    p.q = q; p.k = k; p.v = v; p.mask = mask; p.scale = scale_dev; p.o = out;
    p.bh = (int)batch_heads; p.sq = (int)seq_q; p.sk = (int)seq_k;
    p.mask_group = mask ? (int)mask_group : 1;
    p.scale_is_div = scale_is_div;
    p.scale_imm = scale;
    p.causal = causal;
    p.o_heads = (int)heads;
    p.mask_rows = (mask && mask_2d) ? (int)seq_q : 0;
    const bool bf = dtype == INFINI_DT_BF16;
    // NT = 2 (32 query rows per wave, 128 per workgroup) for both head sizes. At D = 64 that is 153-168 VGPRs = 3 waves per
    // SIMD and, on BERT's 384 x 512 problem, 1536 workgroups = exactly two rounds of the chip's 768 slots; the 64-row
    // variant (256 VGPRs, 2 waves per SIMD, 768 workgroups on 512 slots) measured 52.9 / 62.9 us (plain / masked) against
    // 49.5 / 50.6 us for this one, and 171 vs 162 us at S = 2048.
    if (head_dim == 64)
        return bf ? launch_attn<Bf16Traits, 64, 2>(rt, p) : launch_attn<F16Traits, 64, 2>(rt, p);
    return bf ? launch_attn<Bf16Traits, 128, 2>(rt, p) : launch_attn<F16Traits, 128, 2>(rt, p);
}
