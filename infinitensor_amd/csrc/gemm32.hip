// gemm_fast32: the fp32 MatMul tile kernel for gfx950 — (64 T) x (64 T) x 32 tile, 4 waves (2 x 2, (32 T)^2 each; T = 2: 128^2
// tiles, T = 1: 64^2 tiles for problems that would leave most CUs without a 128^2 tile, e.g. BASELINE config 1's 512^3),
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate: an fma chain per output element), operands staged by
// LDS-DMA (global_load_lds_dwordx4) into a double buffer, one barrier per K-tile, >= two workgroups per CU (64 / 32 KiB LDS).
//
// Replaces the fp32 branch of matmulCublas::do_compute (reference: src/kernels/cuda/matmul.cc:51-64,140-174 — cublasGemmEx
// with CUDA_R_32F / the operator's compute type) for shapes with at least half a tile per CU; everything else (ragged,
// unaligned, transA, tiny) stays on gemm_generic32. `MatmulObj::getComputeType()` "tf32" / "bf16" / "fp16" ask cuBLAS for
// reduced-precision products; this backend always multiplies fp32 operands exactly (a deliberate deviation: more accurate,
// and the 1e-4 gate of north_star is stated for fp32) — DESIGN.md section 4.
//
// Layout. A is K-major ([m][k], row-major A). Its tile image is [128 rows][32 k] with 128-byte rows, the 16-byte chunk
// index XORed with (row >> 1) & 7 — the very geometry of the 16-bit kernels' K-major tile (64 halves = 32 floats), so a
// lane's ds_read_b128 of 4 consecutive k of "its" row is conflict-free. B is either K-major too (transB: ONNX Gemm) or
// N-major ([k][n], ONNX MatMul): image [32 k][128 n] with 512-byte rows read with ds_read_b32 (an fp32 MFMA fragment is ONE
// float per lane — no transposing read is needed), chunk index XORed with ((k >> 2) & 1) << 3 so that the two 32-lane
// halves of a read (k rows 4 apart) hit disjoint banks.
// MFMA operand order. v_mfma_f32_32x32x2 takes for lane l: A[row = l % 32][k = l / 32], i.e. two k per instruction. A lane's
// b128 holds k0 .. k0 + 3 with k0 = 4 (l / 32) inside an 8-k block, so instruction s of a block multiplies k = s (lanes
// 0-31) and k = 4 + s (lanes 32-63) — any pairing is fine as long as both operands use the same one. The operands are
// swapped (D = B^T-fragment x A-fragment) so that a lane ends up with 4 consecutive n of one m row: 16-byte stores.
#include "gemm_common.h"
#include <type_traits>

namespace irocm {
namespace f32k {

constexpr int BK = 32;

// T = tile size in units of 64 rows: the tile's 8-row pieces (1 KiB each) are dealt 2 T per wave
template <int T>
__device__ inline void stage_kmajor(const float *base, long ld, int row0, int rows, int k0, char *lds_tile, int w, int lane,
                                    int kend, const float *zeros) {
#pragma unroll
    for (int i = 0; i < 2 * T; ++i) {
        const int piece = w * 2 * T + i;
        const int r = piece * 8 + (lane >> 3);
        const int c_log = (lane & 7) ^ ((r >> 1) & 7);
        int gr = row0 + r;
        gr = gr < rows ? gr : rows - 1; // rows beyond the matrix re-read its last row; never stored
        const float *src = base + (long)gr * ld + k0 + c_log * 4;
        if (k0 + c_log * 4 >= kend)
            src = zeros;
        __builtin_amdgcn_global_load_lds(IROCM_GLB_PTR(src), IROCM_LDS_PTR(lds_tile + piece * 1024), 16, 0, 0);
    }
}

// N-major tile [32 k][64 T n]: rows of 16 T chunks; a 1 KiB piece covers 4 / T k-rows
template <int T>
__device__ inline void stage_nmajor(const float *base, long ld, int col0, int cols, int k0, char *lds_tile, int w, int lane,
                                    int kend, const float *zeros) {
    constexpr int CH = 16 * T; // 16-byte chunks per row
#pragma unroll
    for (int i = 0; i < 2 * T; ++i) {
        const int piece = w * 2 * T + i;
        const int kr = piece * (64 / CH) + lane / CH;
        const int c_log = (lane % CH) ^ (((kr >> 2) & 1) << 3);
        int gc = col0 + c_log * 4;
        gc = gc <= cols - 4 ? gc : cols - 4;
        const float *src = base + (long)(k0 + kr) * ld + gc;
        if (k0 + kr >= kend)
            src = zeros;
        __builtin_amdgcn_global_load_lds(IROCM_GLB_PTR(src), IROCM_LDS_PTR(lds_tile + piece * 1024), 16, 0, 0);
    }
}

// 4 consecutive k (k0 = blk * 8 + 4 (lane / 32)) of row R0 + lane % 32
__device__ inline f32x4 frag_kmajor(const char *lds_tile, int R0, int blk, int lane) {
    const int r = R0 + (lane & 31);
    const int c = (blk * 2 + (lane >> 5)) ^ ((r >> 1) & 7);
    return *(const f32x4 *)(lds_tile + r * 128 + c * 16);
}
// the same 4 k of column C0 + lane % 32 from the N-major image
template <int T> __device__ inline f32x4 frag_nmajor(const char *lds_tile, int C0, int blk, int lane) {
    const int n = C0 + (lane & 31);
    f32x4 v;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int kr = blk * 8 + (lane >> 5) * 4 + s;
        const int chunk = (n >> 2) ^ (((kr >> 2) & 1) << 3);
        v[s] = *(const float *)(lds_tile + kr * (256 * T) + chunk * 16 + (n & 3) * 4);
    }
    return v;
}

template <int T, bool B_KMAJOR> __global__ __launch_bounds__(256, 2) void gemm_fast32(GemmArgs p) {
    constexpr int BM = 64 * T, BN = 64 * T, WT = 32 * T; // workgroup tile, wave tile
    constexpr int TILE_BYTES = BM * BK * 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto a_tile = [&](int buf) -> char * { return smem + buf * TILE_BYTES; };
    auto b_tile = [&](int buf) -> char * { return smem + (2 + buf) * TILE_BYTES; };
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = w >> 1, wn = w & 1;

    // workgroup -> (batch, tile_m, tile_n): XCD-aware remap, then grouped raster (8 tile-rows), as gemm_fast128
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

    const float *A = (const float *)p.a + (long)ib * p.a_bs;
    const float *B = (const float *)p.b + (long)ib * p.b_bs;
    const long lda = p.a_rs;
    const long ldb = B_KMAJOR ? p.b_cs : p.b_rs;
    const float *Z = (const float *)p.zeros;
    auto stage = [&](int buf, int k0) {
        stage_kmajor<T>(A, lda, m0, p.m, k0, a_tile(buf), w, lane, p.k, Z);
        if constexpr (B_KMAJOR)
            stage_kmajor<T>(B, ldb, n0, p.n, k0, b_tile(buf), w, lane, p.k, Z);
        else
            stage_nmajor<T>(B, ldb, n0, p.n, k0, b_tile(buf), w, lane, p.k, Z);
    };

    f32x16 acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                acc[i][j][e] = 0.f;

    const int nk = (p.k + BK - 1) / BK;
    // One K-tile step with a COMPILE-TIME buffer index (see gemm_fast128: hipcc keeps the next tile's LDS-DMA in flight
    // across the ds_reads of this one only when it can prove the LDS ranges distinct).
    auto step = [&](auto bufc, int kt) {
        constexpr int buf = decltype(bufc)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk)
            stage(buf ^ 1, (kt + 1) * BK);
        const char *at = a_tile(buf), *bt = b_tile(buf);
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            f32x4 af[T], bf[T];
#pragma unroll
            for (int i = 0; i < T; ++i)
                af[i] = frag_kmajor(at, wm * WT + i * 32, blk, lane);
#pragma unroll
            for (int j = 0; j < T; ++j)
                bf[j] = B_KMAJOR ? frag_kmajor(bt, wn * WT + j * 32, blk, lane) : frag_nmajor<T>(bt, wn * WT + j * 32, blk, lane);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < T; ++i)
#pragma unroll
                    for (int j = 0; j < T; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j][s], af[i][s], acc[i][j], 0, 0, 0);
        }
    };
    stage(0, 0);
    for (int kt = 0; kt < nk; kt += 2) {
        step(std::integral_constant<int, 0>{}, kt);
        if (kt + 1 < nk)
            step(std::integral_constant<int, 1>{}, kt + 1);
    }

    // C/D layout of the 32x32 MFMA with swapped operands: lane l holds column m = l % 32 of the (n x m) product; register
    // 4 g + r is row n = 8 g + 4 (l / 32) + r — four consecutive n per g: one 16-byte store
    float *C = (float *)p.c + (long)ib * p.c_bs;
    const float *bias = (const float *)p.bias;
    const bool vec = (m0 + BM <= p.m) && (n0 + BN <= p.n) && (p.n % 4 == 0) && ((((uintptr_t)p.c) & 15) == 0) && p.hs_d == 0;
#pragma unroll
    for (int i = 0; i < T; ++i) {
        const int row = m0 + wm * WT + i * 32 + (lane & 31);
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = n0 + wn * WT + j * 32 + g * 8 + (lane >> 5) * 4;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    v[r] = acc[i][j][g * 4 + r];
                if (vec) {
                    if (bias) {
                        const float *bp = bias + (long)ib * p.bias_b + (long)row * p.bias_m + (long)col * p.bias_n;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            v[r] += bp[(long)r * p.bias_n];
                    }
                    if (p.act) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            v[r] = apply_act(v[r], p.act);
                    }
                    *(f32x4 *)(C + (long)row * p.n + col) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (row < p.m && col + r < p.n) {
                            float x = v[r];
                            if (bias)
                                x += bias[(long)ib * p.bias_b + (long)row * p.bias_m + (long)(col + r) * p.bias_n];
                            C[c_off(p, row, col + r)] = apply_act(x, p.act);
                        }
                }
            }
    }
}


// ---- fp32 Conv2d as an implicit GEMM on the same tile machinery (round 5) ------------------------------------------------------------
// Reference: the fp32 convolution of src/kernels/cuda/conv.cc:57-168 (cuDNN implicit GEMM; fp32 is the dtype north_star's 1e-4 gate and
// the intelcpu baseline are stated in). Until round 5 every fp32 convolution ran on conv_direct32 — one output per thread, a serial
// fma chain: VALU-bound by construction.
// GEMM view (groups == 1): Y[img][f][pix] = sum_k W[f][k] * B[k][q], k = (c R + r) S + s, q = img * OH * OW + pix (columns run across
// images, so 7 x 7 planes do not waste tiles), B[k][q] = X[img][c][oy sh - ph + r dh][ox sw - pw + s dw] or 0 outside the image.
//   A = the FCRS weights as they lie: [F][K] K-major rows — staged by LDS-DMA exactly like gemm_fast32's A (K % 4 == 0).
//   B is gathered: thread t owns ONE column q = n0 + t % BN of the tile and 32 / (256 / BN) k-rows of it, so the column's geometry
//   (image base, iy0, ix0) is three registers for the whole kernel, the k-row's (c, r, s) is WAVE-uniform scalar arithmetic, a wave's
//   loads of one k-row are 64 consecutive pixels (coalesced wherever a row of the plane is), and the LDS writes of a k-row are
//   conflict-free. Register-staged: the loads of K-tile t + 1 are issued before the MFMAs of tile t and written to LDS behind them
//   (64-cycle fp32 MFMAs leave the vector ALU idle 15 cycles of 16: the gather's ~8 instructions per element ride in that shadow).
// Exact fp32 products and sums (v_mfma_f32_32x32x2_f32), summation order over k differs from conv_direct32's chain only by the MFMA's
// two-k interleave: 1e-4 relative is kept with orders of magnitude to spare (tests).
struct Conv32Args {
    const float *x, *w, *bias, *res;
    float *y;
    int nimg, c, h, wd, f, r, s;
    int ph, pw, sh, sw, dh, dw;
    int oh, ow;
    int k;            // K of the GEMM = row pitch of the weight image: c * r * s rounded up to 4 (tap-major: r * s * cp)
    int kreal;        // c * r * s (k-major form: elements at k >= kreal are padding)
    long ncols;       // nimg * oh * ow
    int tiles_m, tiles_n;
    int act;
    unsigned x_bytes; // range of the input's buffer descriptor
    unsigned ohw_m, ow_m, rs_m, s_m; // floor(2^32 / d) of oh * ow, ow, r * s, s
    const void *zeros;
    int cp;           // TAP-MAJOR form: channels per tap in the packed weights [F][R S][cp] (cp = C rounded up to 32; k = tap * cp + c)
    // split-K (layers whose tiles cannot fill the chip: 7 x 7 planes at batch 32 are 200 tiles of 144 K-tiles): `split` workgroups per
    // tile, each over a contiguous range of K-tiles, store RAW fp32 sums into partial[slice][N][F][OH][OW]; conv32_splitk_reduce adds the
    // slices in ascending order (bit-reproducible), the bias, the residual and the activation. split == 1: none of this.
    int split;
    float *partial;
    long out_elems;   // N F OH OW
    unsigned long long *trace; // TRACE instantiation only (tools/conv32_timeline.py): [grid][4 waves][kC32Slots] s_memtime stamps
};
constexpr int kC32Slots = 128;

// y = act(sum over slices of partial + bias[f] + res); V consecutive elements per thread (V = 4 when planes are multiples of 4: one filter)
template <int V>
__global__ __launch_bounds__(256) void conv32_splitk_reduce(const float *__restrict__ partial, int split, long out_elems, const float *__restrict__ bias,
                                                            const float *__restrict__ res, float *__restrict__ y, int f, int ohw, unsigned ohw_m,
                                                            unsigned f_m, int act) {
    const long nvec = out_elems / V;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const long e = i * V;
        float v[V];
#pragma unroll
        for (int r = 0; r < V; ++r)
            v[r] = 0.f;
        for (int sl = 0; sl < split; ++sl) {
            if constexpr (V == 4) {
                const f32x4 t = *(const f32x4 *)(partial + (long)sl * out_elems + e);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    v[r] += t[r];
            } else {
                v[0] += partial[(long)sl * out_elems + e];
            }
        }
        unsigned plane, pix, img, ff;
        udivmod_m((unsigned)e, (unsigned)ohw, ohw_m, plane, pix);
        udivmod_m(plane, (unsigned)f, f_m, img, ff);
        const float bv = bias ? bias[ff] : 0.f;
#pragma unroll
        for (int r = 0; r < V; ++r) {
            float x = v[r] + bv;
            if (res)
                x += res[e + r];
            v[r] = apply_act(x, act);
        }
        if constexpr (V == 4)
            *(f32x4 *)(y + e) = f32x4{v[0], v[1], v[2], v[3]};
        else
            y[e] = v[0];
    }
}


// FCRS -> [F][R S][cp] (zero-filled above C): the tap-major weight image of conv_igemm32<T, true>
__global__ __launch_bounds__(256) void conv_repack_w32(const float *__restrict__ w, float *__restrict__ o, int f, int c, int rs, int cp) {
    const long total = (long)f * rs * cp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cc = (int)(i % cp);
        const long q = i / cp;
        const int t = (int)(q % rs);
        const int ff = (int)(q / rs);
        o[i] = cc < c ? w[((long)ff * c + cc) * rs + t] : 0.f;
    }
}

// TM (tap-major K): k = tap * cp + c over the re-packed weights — a K-tile is 32 channels of ONE tap, so the tap's (dy, dx), its bounds
// test and the column's base offset are computed once per K-tile and an element costs one add + one select + its load. With k =
// (c R + r) S + s over the FCRS weights as they lie (TM = false: layers with fewer than 32 channels) every element decodes its own
// (c, r, s): the PMC pass of the first version showed 2.5 scalar instructions per vector one — 370 per K-tile and wave beside 16 MFMAs —
// and the layers at 0.16-0.35 of the fp32 peak.
template <int T, bool TM, bool TRACE = false> __global__ __launch_bounds__(256, 2) void conv_igemm32(Conv32Args p) {
    constexpr int BM = 64 * T, BN = 64 * T, WT = 32 * T;
    constexpr int TILE_BYTES = BM * BK * 4;
    constexpr int KR_STEP = 256 / BN;      // k-rows covered by the 256 threads at once (2 for T = 2, 4 for T = 1)
    constexpr int NE = BK / KR_STEP;       // elements per thread and K-tile (16 / 8)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto a_tile = [&](int buf) -> char * { return smem + buf * TILE_BYTES; };
    auto b_tile = [&](int buf) -> char * { return smem + (2 + buf) * TILE_BYTES; };
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = w >> 1, wn = w & 1;
    // TRACE (round 6; tools/conv32_timeline.py): every wave stamps s_memtime at four points of every step into an LDS strip behind the
    // tiles (no VMEM traffic: the counted waits are untouched) and dumps it at the end
    int tslot = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if constexpr (TRACE) {
            const unsigned long long tm_ = __builtin_amdgcn_s_memtime();
            if (tslot < kC32Slots && lane == 0)
                *(unsigned long long *)(smem + 4 * TILE_BYTES + (w * kC32Slots + tslot) * 8) = tm_;
            ++tslot;
        }
    };
    const int S = p.split;
    const int slice = S > 1 ? (int)(blockIdx.x % (unsigned)S) : 0;
    unsigned wg = S > 1 ? xcd_remap(blockIdx.x / (unsigned)S, gridDim.x / (unsigned)S) : xcd_remap(blockIdx.x, gridDim.x);
    constexpr int GROUP_M = 8;
    const unsigned per_group = GROUP_M * p.tiles_n;
    const unsigned group = wg / per_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int tm = first_m + (wg % per_group) % gsz;
    const int tn = (wg % per_group) / gsz;
    const int m0 = tm * BM;
    const long n0 = (long)tn * BN;
    // this workgroup's K-tiles [kt_begin, kt_end): all of them, or an EVEN number per slice (steps come in pairs; a pair that runs past
    // the real K multiplies zero weights, but a pair must never run into the next slice's tiles)
    const int nk_all = (p.k + BK - 1) / BK;
    const int nk_s = S > 1 ? (((nk_all + S - 1) / S + 1) & ~1) : nk_all;
    const int kt_begin = slice * nk_s, kt_end = kt_begin + nk_s;

    // this thread's column: output pixel q -> (image, oy, ox)
    const int ncol = t % BN;
    const int kr0 = __builtin_amdgcn_readfirstlane(t / BN); // (wave-uniform: BN >= 64)
    long q = n0 + ncol;
    const bool qlive = q < p.ncols;
    if (!qlive)
        q = p.ncols - 1;
    unsigned img, pix, oy, ox;
    udivmod_m((unsigned)q, (unsigned)(p.oh * p.ow), p.ohw_m, img, pix);
    udivmod_m(pix, (unsigned)p.ow, p.ow_m, oy, ox);
    const int iy0 = (int)oy * p.sh - p.ph, ix0 = (int)ox * p.sw - p.pw;
    const int pix_base = (int)(((long)img * p.c * p.h + iy0) * p.wd + ix0); // element index of tap (c 0, r 0, s 0); may be negative
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const int hw = p.h * p.wd;
    const unsigned lds_col = (unsigned)((ncol & 3) * 4); // byte inside the 16-byte chunk; chunk index depends on the k-row

    float sv_a[NE], sv_b[NE]; // two register stages of gathered elements (see step below)
    // tap-major state of the K-tile being gathered: tap (tr, ts), channel block c0; per thread: base offset and validity of that tap
    int g_tr = 0, g_ts = 0, g_c0 = 0, g_base = 0;
    bool g_ok = false;
    auto tap_setup = [&]() __attribute__((always_inline)) {
        const int dy = g_tr * p.dh, dx = g_ts * p.dw;
        // (bitwise &: with && hipcc built a divergent branch around the second pair of compares — a block boundary in the middle of the
        // K loop, behind which its wait insertion no longer counts the requests in flight)
        g_ok = qlive & (g_tr < p.r) & ((unsigned)(iy0 + dy) < (unsigned)p.h) & ((unsigned)(ix0 + dx) < (unsigned)p.wd);
        g_base = pix_base + dy * p.wd + dx + (g_c0 + kr0) * hw;
    };
    auto tap_advance = [&]() __attribute__((always_inline)) { // the next K-tile: 32 more channels, then the next tap
        g_c0 += BK;
        if (g_c0 >= p.cp) {
            g_c0 = 0;
            g_ts = g_ts + 1 == p.s ? 0 : g_ts + 1;
            g_tr += g_ts == 0 ? 1 : 0;
        }
    };
    auto gather_one = [&](int k0, int i) __attribute__((always_inline)) { // requests element i of K-tile k0 .. k0 + 31 of this column
        if constexpr (TM) {
            const bool ok = g_ok & (g_c0 + kr0 + i * KR_STEP < p.c);
            return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, ok ? (g_base + i * (KR_STEP * hw)) * 4 : (int)0x7ffffff0, 0, 0));
        } else {
            const int k = k0 + kr0 + i * KR_STEP; // wave-uniform
            unsigned cc, rs, rr, ss;
            udivmod_m((unsigned)k, (unsigned)(p.r * p.s), p.rs_m, cc, rs);
            udivmod_m(rs, (unsigned)p.s, p.s_m, rr, ss);
            const int dy = (int)rr * p.dh, dx = (int)ss * p.dw;
            const bool ok = qlive & (k < p.kreal) & ((unsigned)(iy0 + dy) < (unsigned)p.h) & ((unsigned)(ix0 + dx) < (unsigned)p.wd);
            const int off = pix_base + (int)cc * hw + dy * p.wd + dx;
            return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, ok ? off * 4 : (int)0x7ffffff0, 0, 0));
        }
    };
    auto gather = [&](float (&dst)[NE], int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NE; ++i)
            dst[i] = gather_one(k0, i);
    };
    // into the N-major image gemm_fast32's fragments read. Written as inline-asm ds_writes: hipcc orders every LDS access it can see
    // behind the LDS-DMA in flight (it cannot tell that the DMA fills the OTHER operand's tile) — with plain stores the second step
    // of the loop body waited vmcnt(0) here, i.e. for the elements requested a moment ago. What the asm leaves to the compiler is the
    // register dependency on the loads of `src`, which it counts exactly; completion (lgkmcnt) is the next step's barrier wait.
    const unsigned b_lds0 = (unsigned)(unsigned long)IROCM_LDS_PTR(smem) + 2 * TILE_BYTES;
    auto scatter = [&](int buf, const float (&src)[NE]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int kr = kr0 + i * KR_STEP;
            const unsigned chunk = (unsigned)(ncol >> 2) ^ ((unsigned)((kr >> 2) & 1) << 3);
            const unsigned addr = b_lds0 + (unsigned)buf * TILE_BYTES + (unsigned)kr * (256 * T) + chunk * 16 + lds_col;
            asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(src[i]) : "memory");
        }
    };

    f32x16 acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                acc[i][j][e] = 0.f;

    const float *Z = (const float *)p.zeros;
    // Pipeline (second version, round 5): the gathered elements pass through TWO register stages. Step kt multiplies tile kt out of
    // LDS buffer kt % 2, REQUESTS the elements of tile kt + 2 between its MFMAs and, behind them, writes tile kt + 1 — requested a whole
    // step earlier — into the other buffer. (The first version requested tile kt + 1 in step kt and wrote it at the end of the same
    // step: `s_waitcnt vmcnt(0)` right behind the last request, one full memory round trip exposed per K-tile; MFMA busy 0.37.)
    // Waits at the top of a step are counted: in issue order the outstanding requests are [weights of tile kt] [NE elements of tile
    // kt + 1], so vmcnt(NE) is "the weights have landed" (hipcc's __syncthreads would drain everything).
    // prologue: tile 0's weights by LDS-DMA, its gathered column through registers; tile 1's elements requested
    stage_kmajor<T>(p.w, p.k, m0, p.f, kt_begin * BK, a_tile(0), w, lane, p.k, Z);
    if constexpr (TM) {
        if (kt_begin) { // (a later slice: which tap and channel block its first K-tile is — scalar, once)
            const int per_tap = p.cp / BK, tap = kt_begin / per_tap;
            g_c0 = (kt_begin - tap * per_tap) * BK;
            g_tr = tap / p.s;
            g_ts = tap - g_tr * p.s;
        }
        tap_setup();
    }
    gather(sv_a, kt_begin * BK);
    scatter(0, sv_a); // (hipcc waits for the loads here)
    if constexpr (TM) {
        tap_advance();
        tap_setup();
    }
    gather(sv_b, (kt_begin + 1) * BK); // (past the last tile: every element out of range — zeros, never multiplied)
    auto step = [&](auto bufc, int kt, float (&cur)[NE], float (&nxt)[NE]) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        stamp(); // (TRACE: step entry)
        // tile kt's weights have landed; every thread's column of tile kt is in LDS (its ds_writes drained) — then the barrier
        if constexpr (TRACE) { // (the same wait in two parts, a stamp between them: what the counted wait costs, what the barrier costs)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NE) : "memory");
            stamp();
            asm volatile("s_barrier" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NE) : "memory");
        }
        stamp(); // (TRACE: behind the barrier)
        // (unconditional — past the last tile the weights come from the zero block and every gathered element is out of range — so
        // that the gather sits in the SAME basic block as the MFMAs and can be scheduled between them)
        stage_kmajor<T>(p.w, p.k, m0, p.f, (kt + 1) * BK, a_tile(buf ^ 1), w, lane, p.k, Z);
        if constexpr (TM) {
            tap_advance(); // -> tile kt + 2
            tap_setup();
        }
        const char *at = a_tile(buf), *bt = b_tile(buf);
        // The gather rides BETWEEN this tile's MFMAs, one element (the bounds tests, one load: ~25 instructions in the k-major form)
        // behind every MFMA group: a 64-cycle fp32 MFMA leaves ~12 vector-ALU issue slots before the next one can start, but only
        // instructions that sit between two MFMAs in program order can use them. Left alone hipcc issues the whole gather first
        // (~2 k cycles in front of 4 k cycles of MFMAs: 0.18-0.35 of the fp32 peak at batch 32), and sched_group_barrier did not move it;
        // the order is pinned with scheduling fences instead.
        constexpr int PER = (16 * T * T) / NE; // MFMAs per gathered element: 4
        int mi = 0;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            f32x4 af[T], bf[T];
#pragma unroll
            for (int i = 0; i < T; ++i)
                af[i] = frag_kmajor(at, wm * WT + i * 32, blk, lane);
#pragma unroll
            for (int j = 0; j < T; ++j)
                bf[j] = frag_nmajor<T>(bt, wn * WT + j * 32, blk, lane);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < T; ++i)
#pragma unroll
                    for (int j = 0; j < T; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j][s], af[i][s], acc[i][j], 0, 0, 0);
                        if (++mi % PER == 0) {
                            __builtin_amdgcn_sched_barrier(0);
                            nxt[mi / PER - 1] = gather_one((kt + 2) * BK, mi / PER - 1);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
        }
        __builtin_amdgcn_sched_barrier(0);
        stamp(); // (TRACE: the 16 T^2 MFMAs and the NE requests are issued)
        scatter(buf ^ 1, cur); // tile kt + 1, requested one step ago (buffer buf ^ 1 was last read in step kt - 1, behind this step's barrier)
    };
    // Steps come in pairs, unconditionally: with an odd tile count the last step multiplies a tile of zero weights by out-of-range
    // (zero) elements. (An exit between the two made hipcc rotate the loop around the second step.) What the ISA shows: hipcc does not
    // count requests across the LDS-DMA — it puts vmcnt(0) in front of the FIRST step's LDS writes (behind that step's own requests)
    // and, everything having landed, nothing in front of the second step's: one exposed round trip per two K-tiles instead of one per
    // K-tile. (Four steps per iteration gave vmcnt(0) in front of the first and third write blocks — the same ratio at 208 VGPRs.)
    for (int kt = kt_begin; kt < kt_end; kt += 2) {
        step(std::integral_constant<int, 0>{}, kt, sv_b, sv_a);
        step(std::integral_constant<int, 1>{}, kt + 1, sv_a, sv_b);
    }

    if constexpr (TRACE) {
        stamp();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        for (int i = lane; i < kC32Slots; i += 64)
            p.trace[((size_t)blockIdx.x * 4 + w) * kC32Slots + i] =
                i < tslot ? *(unsigned long long *)(smem + 4 * TILE_BYTES + (w * kC32Slots + i) * 8) : 0ull;
    }
    // epilogue: lane l holds filter m = l % 32 and, per g, four consecutive columns q. A split-K slice stores its raw sums into its
    // plane of `partial` (same NCHW indexing); bias / residual / activation then belong to conv32_splitk_reduce.
    const int ohw = p.oh * p.ow;
    const bool fin = S == 1;
    float *const Y = fin ? p.y : p.partial + (long)slice * p.out_elems;
    const float *const R = fin ? p.res : nullptr;
    const int act = fin ? p.act : 0;
#pragma unroll
    for (int i = 0; i < T; ++i) {
        const int frow = m0 + wm * WT + i * 32 + (lane & 31);
        const float bv = (fin && p.bias && frow < p.f) ? p.bias[frow] : 0.f;
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const long qc = n0 + wn * WT + j * 32 + g * 8 + (lane >> 5) * 4;
                if (frow >= p.f || qc >= p.ncols)
                    continue;
                unsigned im2, px2;
                udivmod_m((unsigned)qc, (unsigned)ohw, p.ohw_m, im2, px2);
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    v[r] = acc[i][j][g * 4 + r] + bv;
                const long o = ((long)im2 * p.f + frow) * ohw + px2;
                if (px2 + 4 <= (unsigned)ohw && qc + 4 <= p.ncols && (o & 3) == 0 && ((((uintptr_t)Y) & 15) == 0)) {
                    if (R) {
                        const f32x4 rv = *(const f32x4 *)(R + o);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            v[r] += rv[r];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        v[r] = apply_act(v[r], act);
                    *(f32x4 *)(Y + o) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const long qq = qc + r;
                        if (qq >= p.ncols)
                            break;
                        unsigned im3, px3;
                        udivmod_m((unsigned)qq, (unsigned)ohw, p.ohw_m, im3, px3);
                        const long oo = ((long)im3 * p.f + frow) * ohw + px3;
                        float x = v[r];
                        if (R)
                            x += R[oo];
                        Y[oo] = apply_act(x, act);
                    }
                }
            }
    }
}

} // namespace f32k

static bool al16p(const void *p) { return (((uintptr_t)p) & 15) == 0; }

// A K-major with 16-byte rows; B K-major or N-major with 16-byte rows; K a multiple of 4 (a K tail inside the last
// 32-wide tile is zero-filled); the generic kernel serves the rest
bool fast32_supported(const GemmArgs &p, bool akm, bool bkm) {
    if (!akm || p.k % 4 != 0 || p.k < 4 || p.m < 1 || p.n < 4)
        return false;
    if (!al16p(p.a) || !al16p(p.b) || (p.a_bs % 4) || (p.b_bs % 4) || (p.a_rs % 4))
        return false;
    if (bkm ? (p.b_cs % 4 != 0) : (p.b_rs % 4 != 0 || p.n % 4 != 0))
        return false;
    return true;
}

// small != 0: 64^2 tiles (a problem with fewer than ~half a 128^2 tile per CU)
int launch_fast32(infiniRocmRuntime_t rt, GemmArgs p, bool bkm, int small) {
    const int bm = small ? 64 : 128;
    p.tiles_m = (int)ceil_div(p.m, bm);
    p.tiles_n = (int)ceil_div(p.n, bm);
    const long total = (long)p.tiles_m * p.tiles_n * p.batch;
    if (total >= (1l << 31))
        IROCM_FAIL(INFINI_ROCM_INVALID_ARGUMENT, "matmul: too many tiles");
    const unsigned grid = (unsigned)total;
    const size_t lds = 4 * (size_t)bm * f32k::BK * 4;
#define IROCM_F32K(T_, BK_)                                                                        \
    do {                                                                                           \
        auto kern = f32k::gemm_fast32<T_, BK_>;                                                    \
        IROCM_LDS_ATTR(kern, (int)lds, rt);                                                        \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, rt->stream, p);                       \
    } while (0)
    if (small) {
        if (bkm) IROCM_F32K(1, true); else IROCM_F32K(1, false);
    } else {
        if (bkm) IROCM_F32K(2, true); else IROCM_F32K(2, false);
    }
#undef IROCM_F32K
    IROCM_LAUNCH_CHECK("gemm_fast32");
    return INFINI_ROCM_OK;
}


// fp32 Conv2d (groups == 1) as an implicit GEMM on v_mfma_f32_32x32x2_f32. Returns -1 when the operands do not qualify
// (the caller keeps conv_direct32), a status otherwise.
int launch_conv_igemm32(infiniRocmRuntime_t rt, const void *x, const void *w, const void *bias, const void *res, void *y, int64_t n,
                        int64_t c, int64_t h, int64_t wd, int64_t f, int r, int s, int ph, int pw, int sh, int sw, int dh, int dw, int oh,
                        int ow, int act) {
    const int64_t ncols = n * oh * ow;
    const bool tm = c >= 32; // tap-major K over re-packed weights (see the kernel); layers with fewer channels decode k per element
    const int64_t cp = tm ? (c + 31) / 32 * 32 : c;
    // (k-major rows whose length is not a multiple of 4 — the 3-channel 7 x 7 stem: 147 — are copied into zero-padded 16-byte rows)
    const bool padk = !tm && (c * r * s) % 4 != 0;
    const int64_t k = tm ? cp * r * s : (c * r * s + 3) / 4 * 4;
    if (k < 4 || (!tm && !padk && !al16p(w)) || (((uintptr_t)w) & 3) != 0 || (((uintptr_t)x) & 3) != 0 || (((uintptr_t)y) & 3) != 0 || (res && (((uintptr_t)res) & 3) != 0))
        return -1;
    if (n * c * h * wd * 4 >= (1ll << 31) - 64 || ncols >= (1ll << 31) - 256 || n * f * oh * ow >= (1ll << 31) || k >= (1ll << 24))
        return -1;
    f32k::Conv32Args p;
    p.x = (const float *)x; p.w = (const float *)w; p.bias = (const float *)bias; p.res = (const float *)res; p.y = (float *)y;
    // Tile size, measured on ResNet-50's layers at batch 32 (tools/conv32_bench.py --forms, us with 64^2 / 128^2 tiles): C128 28^2 3 x 3
    // 97.9 / 115.6; C128 56^2 3 x 3 / 2 99.0 / 127.7; C256 56^2 -> 128 1 x 1 (784 tiles of 128^2 = 3 per CU) 82.7 / 118.7; C256 14^2 3 x 3
    // 109 / 217: the 64^2 tiles (83 registers, 32 KB of LDS: five workgroups per CU hide the gather's latency) win everywhere there, so
    // the 128^2 form (182 registers, two workgroups per CU) is kept for problems with at least eight of its tiles per CU only.
    bool small = f <= 64 || ceil_div(f, 128) * ceil_div(ncols, 128) < 8 * (int64_t)rt->num_cu;
    if (const char *e = getenv("IROCM_CONV32_TILE")) // measurement hook (tools/conv32_bench.py --forms): 1 = 64^2 tiles, 2 = 128^2
        small = atoi(e) == 1 ? true : (atoi(e) == 2 ? false : small);
    const int bm = small ? 64 : 128;
    const int64_t tiles = ceil_div(f, bm) * ceil_div(ncols, bm);
    // Split-K: the 64^2 tiles of a 7 x 7-plane layer at batch 32 number 200 (one per CU, no partner to hide the gather behind) with
    // 64-144 K-tiles each. `split` workgroups per tile over even runs of K-tiles, raw sums to the workspace, one reduce pass. Measured
    // (tools/conv32_bench.py --forms, us with 1 / 2 / 4 slices): 200 tiles — C512 7^2 3 x 3 150 / 120 / 116, C512 14^2 3 x 3 / 2 154 / 121 /
    // 118, C2048 -> 512 7^2 77 / 63 / 63; 392 tiles — C256 14^2 3 x 3 (72 K-tiles) 108 / 107 / 100, C256 28^2 3 x 3 / 2 109 / 107 / 101,
    // C1024 -> 256 14^2 (32 K-tiles) 52 / 55 / 55; 800 tiles: slower with any split. Hence four slices up to one tile per CU, and up to
    // two tiles per CU when K is long enough to pay for the second pass.
    const int64_t nk_all = ceil_div(k, f32k::BK);
    int split = 1;
    if (small && ((tiles <= rt->num_cu && nk_all >= 32) || (tiles <= 2 * (int64_t)rt->num_cu && nk_all >= 64)))
        split = 4;
    if (const char *e = getenv("IROCM_CONV32_SPLIT")) // measurement / test hook: 1 = never, 2 / 4 = that factor wherever a slice keeps >= 2 K-tiles
        if (small && atoi(e) >= 1 && atoi(e) <= 8 && nk_all >= 2 * atoi(e))
            split = atoi(e);
    const int64_t out_elems = n * f * oh * ow;
    const size_t partial_bytes = split > 1 ? (size_t)split * out_elems * 4 : 0;
    float *partial = nullptr;
    // The caller may have parked THIS layer's input in the workspace (the plugin's planner bridges a conv's input there when the fused
    // chain's output was planned onto it): whatever this launcher takes from the workspace goes behind the input's last byte.
    size_t front = 0;
    if (rt->workspace && (const char *)x >= (const char *)rt->workspace && (const char *)x < (const char *)rt->workspace + rt->workspace_bytes)
        front = (((size_t)((const char *)x - (const char *)rt->workspace) + (size_t)(n * c * h * wd * 4)) + 255) & ~(size_t)255;
    if (tm || padk) {
        const int kind = tm ? 2 : 3;
        // the tap-major (or row-padded) weight image: cached per graph weight when the caller declared the weights constant (the plugin does), else
        // rebuilt per call in the workspace (as conv_s1.hip does for the 16-bit kernels' [RS][F][C] image)
        const size_t w_bytes = ((size_t)f * k * 4 + 255) & ~(size_t)255;
        const void *packed = nullptr;
        hipStream_t pack_stream = rt->stream;
        bool need_pack = true;
        const bool cached = rt->conv_const_weights != 0;
        if (cached) {
            packed = wcache_lookup(rt, w, (int)f, (int)c, r * s, kind);
            if (packed) {
                need_pack = false;
            } else {
                void *buf = nullptr;
                const int st = wcache_insert(rt, w, (size_t)f * c * r * s * 4, (int)f, (int)c, r * s, kind, w_bytes, &buf, &pack_stream);
                if (st != INFINI_ROCM_OK)
                    return st;
                packed = buf;
            }
        } else {
            void *ws = nullptr;
            const int st = infini_rocm_workspace(rt, front + w_bytes + partial_bytes, &ws);
            if (st != INFINI_ROCM_OK)
                return st;
            packed = (char *)ws + front;
            if (partial_bytes)
                partial = (float *)((char *)ws + front + w_bytes);
        }
        if (need_pack) {
            long g = ceil_div((long)f * k, 256);
            if (g > 4096) g = 4096;
            // (the row-padded form is the same copy with the whole row as one "tap" of c r s "channels")
            hipLaunchKernelGGL(f32k::conv_repack_w32, dim3((unsigned)g), dim3(256), 0, pack_stream, (const float *)w, (float *)const_cast<void *>(packed),
                               (int)f, tm ? (int)c : (int)(c * r * s), tm ? r * s : 1, tm ? (int)cp : (int)k);
            if (hipError_t e = hipGetLastError(); e != hipSuccess) {
                if (cached)
                    wcache_forget(rt, packed);
                IROCM_FAIL(INFINI_ROCM_HIP_ERROR, "launch of conv_repack_w32 failed: %s", hipGetErrorString(e));
            }
            if (cached) {
                const int st = wcache_commit(rt, pack_stream);
                if (st != INFINI_ROCM_OK) {
                    wcache_forget(rt, packed);
                    return st;
                }
            }
        }
        p.w = (const float *)packed;
    }
    if (partial_bytes && !partial) { // (weights cached or used as they lie: the workspace is all the slices')
        void *ws = nullptr;
        const int st = infini_rocm_workspace(rt, front + partial_bytes, &ws);
        if (st != INFINI_ROCM_OK)
            return st;
        partial = (float *)((char *)ws + front);
    }
    p.split = split; p.partial = partial; p.out_elems = out_elems;
    p.nimg = (int)n; p.c = (int)c; p.h = (int)h; p.wd = (int)wd; p.f = (int)f; p.r = r; p.s = s;
    p.ph = ph; p.pw = pw; p.sh = sh; p.sw = sw; p.dh = dh; p.dw = dw;
    p.oh = oh; p.ow = ow;
    p.k = (int)k;
    p.kreal = (int)(c * r * s);
    p.cp = (int)cp;
    p.ncols = ncols;
    p.act = act;
    p.x_bytes = (unsigned)(n * c * h * wd * 4);
    p.ohw_m = udiv_magic((unsigned long long)oh * ow);
    p.ow_m = udiv_magic((unsigned long long)ow);
    p.rs_m = udiv_magic((unsigned long long)r * s);
    p.s_m = udiv_magic((unsigned long long)s);
    p.zeros = rt->zeros;
    p.tiles_m = (int)ceil_div(f, bm);
    p.tiles_n = (int)ceil_div(ncols, bm);
    const long total = (long)p.tiles_m * p.tiles_n;
    if (total >= (1l << 31))
        return -1;
    const size_t lds = 4 * (size_t)bm * f32k::BK * 4;
    p.trace = nullptr;
    // timeline build (tools/conv32_timeline.py): IROCM_CONV32_TRACE = device address (hex) of [grid][4][128] uint64 stamps; 64 x 64 tap-major form
    if (const char *tr = getenv("IROCM_CONV32_TRACE")) {
        p.trace = (unsigned long long *)strtoull(tr, nullptr, 16);
        if (p.trace && small && tm) {
            auto kern = f32k::conv_igemm32<1, true, true>;
            const size_t lds_t = lds + 4 * f32k::kC32Slots * 8;
            IROCM_LDS_ATTR(kern, (int)lds_t, rt);
            hipLaunchKernelGGL(kern, dim3((unsigned)(total * split)), dim3(256), lds_t, rt->stream, p);
            IROCM_LAUNCH_CHECK("conv_igemm32(trace)");
            return INFINI_ROCM_OK; // (timing only: a split launch's reduce pass is not run)
        }
    }
#define IROCM_C32(T_, TM_)                                                                         \
    do {                                                                                           \
        auto kern = f32k::conv_igemm32<T_, TM_>;                                                   \
        IROCM_LDS_ATTR(kern, (int)lds, rt);                                                        \
        hipLaunchKernelGGL(kern, dim3((unsigned)(total * split)), dim3(256), lds, rt->stream, p);  \
    } while (0)
    if (small) {
        if (tm) IROCM_C32(1, true); else IROCM_C32(1, false);
    } else {
        if (tm) IROCM_C32(2, true); else IROCM_C32(2, false);
    }
#undef IROCM_C32
    IROCM_LAUNCH_CHECK("conv_igemm32");
    if (split > 1) {
        rt->last_conv_route = "igemm32_splitk";
        const int ohw = oh * ow;
        const bool v4 = ohw % 4 == 0 && ((((uintptr_t)y) | ((uintptr_t)partial) | (res ? (uintptr_t)res : 0)) & 15) == 0;
        long g = ceil_div(out_elems / (v4 ? 4 : 1), 256);
        if (g > (long)rt->num_cu * 16) g = (long)rt->num_cu * 16;
        if (v4)
            hipLaunchKernelGGL(f32k::conv32_splitk_reduce<4>, dim3((unsigned)g), dim3(256), 0, rt->stream, partial, split, (long)out_elems,
                               (const float *)bias, (const float *)res, (float *)y, (int)f, ohw, p.ohw_m, udiv_magic((unsigned long long)f), act);
        else
            hipLaunchKernelGGL(f32k::conv32_splitk_reduce<1>, dim3((unsigned)g), dim3(256), 0, rt->stream, partial, split, (long)out_elems,
                               (const float *)bias, (const float *)res, (float *)y, (int)f, ohw, p.ohw_m, udiv_magic((unsigned long long)f), act);
        IROCM_LAUNCH_CHECK("conv32_splitk_reduce");
    }
    return INFINI_ROCM_OK;
}

} // namespace irocm
