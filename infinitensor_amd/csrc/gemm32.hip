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

} // namespace irocm
