// MatMul for gfx950: hand-written MFMA GEMM kernels behind infini_rocm_matmul.
//
// Replaces matmulCublas::do_compute (reference: src/kernels/cuda/matmul.cc:67-174), which
// forwards to cublasGemmEx / cublasGemmStridedBatchedEx. Semantics (batch broadcast by zero
// stride, transA/transB, bias broadcast into C) follow that file and the op definition
// src/operators/matmul.cc:26-49.
//
// Kernels
//   gemm_generic16 / gemm_generic32 : any shape / stride / alignment; 64x64 tile, LDS staged
//       through registers with bounds checks, v_mfma_f32_16x16x32_{bf16,f16} or the exact-f32
//       v_mfma_f32_16x16x4_f32. Correctness anchor and fallback for ragged shapes.
//   gemm_fast128<A_KMAJOR,B_KMAJOR> : 16-bit, 128x128x64 tile, 4 waves (2x2, 64x64 each),
//       global->LDS by LDS-DMA (global_load_lds_dwordx4) into a double buffer, one barrier per
//       K-tile. K-major operands ([rows][k], k contiguous) are read with ds_read_b128 from an
//       XOR-swizzled image (swizzle applied on the global SOURCE address because the DMA
//       destination is lane-linear); M/N-major operands ([k][cols], the ONNX MatMul "NN" B and
//       the transA A) are read with the gfx950 transpose read ds_read_b64_tr_b16, so no
//       transposition pass is ever materialised.
//   gemm_256 (gemm256.hip)         : 256x256x64 tile, 8 waves, staggered LOAD | COMPUTE schedule; one tile per
//       workgroup, and the split-K form for few-tile / long-K shapes.
//   gemm256p (gemm256p_kernel.h)   : the same inner loop as ONE persistent workgroup per CU walking its tiles through a
//       flat K-tile pipeline (no cold prologue after the first tile, epilogues overlapped); tile widths 256 / 192 / 128.
#include "gemm_common.h"
#include <type_traits>

namespace irocm {



// ------------------------------------------------------------------------------------------------
// Generic 16-bit kernel
// ------------------------------------------------------------------------------------------------
template <typename Tr> __global__ __launch_bounds__(256) void gemm_generic16(GemmArgs p) {
    constexpr int BM = 64, BN = 64, BK = 32, PITCH = BK + 8;
    __shared__ __attribute__((aligned(16))) unsigned short As[BM][PITCH];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[BN][PITCH];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int ib = blockIdx.z;
    const int m0 = (blockIdx.x / p.tiles_n) * BM, n0 = (blockIdx.x % p.tiles_n) * BN;
    const unsigned short *A = (const unsigned short *)p.a + (long)ib * p.a_bs;
    const unsigned short *B = (const unsigned short *)p.b + (long)ib * p.b_bs;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool a_kc = (p.a_cs == 1), b_kc = (p.b_rs == 1);
    for (int k0 = 0; k0 < p.k; k0 += BK) {
#pragma unroll
        for (int j = 0; j < (BM * BK) / 256; ++j) {
            const int idx = t + j * 256;
            int i, kk;
            if (a_kc) { kk = idx % BK; i = idx / BK; } else { i = idx % BM; kk = idx / BM; }
            const int gi = m0 + i, gk = k0 + kk;
            unsigned short v = 0;
            if (gi < p.m && gk < p.k)
                v = A[(long)gi * p.a_rs + (long)gk * p.a_cs];
            As[i][kk] = v;
        }
#pragma unroll
        for (int j = 0; j < (BN * BK) / 256; ++j) {
            const int idx = t + j * 256;
            int jn, kk;
            if (b_kc) { kk = idx % BK; jn = idx / BK; } else { jn = idx % BN; kk = idx / BN; }
            const int gj = n0 + jn, gk = k0 + kk;
            unsigned short v = 0;
            if (gj < p.n && gk < p.k)
                v = B[(long)gk * p.b_rs + (long)gj * p.b_cs];
            Bs[jn][kk] = v;
        }
        __syncthreads();
        s16x8_t af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
            af[i] = *(const s16x8_t *)&As[wm * 32 + i * 16 + (lane & 15)][(lane >> 4) * 8];
#pragma unroll
        for (int j = 0; j < 2; ++j)
            bf[j] = *(const s16x8_t *)&Bs[wn * 32 + j * 16 + (lane & 15)][(lane >> 4) * 8];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = Tr::mfma(af[i], bf[j], acc[i][j]);
        __syncthreads();
    }
    // one copy of the epilogue per activation class (0 none, 1 relu, 5 A-S Gelu, -1 = p.act at run time): see gemm256p_kernel.h
    auto epilogue = [&](auto actc) __attribute__((always_inline)) {
        constexpr int ACT = decltype(actc)::value;
        auto act1 = [&](float v) {
            if constexpr (ACT == 1) return v > 0.f ? v : 0.f;
            else if constexpr (ACT == 5) return gelu_poly(v);
            else if constexpr (ACT < 0) return apply_act(v, p.act);
            else return v;
        };
        // C/D layout of 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
        unsigned short *C = (unsigned short *)p.c + (long)ib * p.c_bs;
        const unsigned short *bias = (const unsigned short *)p.bias;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
                    const int col = n0 + wn * 32 + j * 16 + (lane & 15);
                    if (row < p.m && col < p.n) {
                        float v = acc[i][j][r];
                        if (bias)
                            v += Tr::to_f32(bias[(long)ib * p.bias_b + (long)row * p.bias_m + (long)col * p.bias_n]);
                        v = act1(v);
                        C[c_off(p, row, col)] = Tr::from_f32(v);
                    }
                }
    };
    if (p.act == 0) epilogue(std::integral_constant<int, 0>{});
    else if (p.act == 1) epilogue(std::integral_constant<int, 1>{});
    else if (p.act == 5) epilogue(std::integral_constant<int, 5>{});
    else epilogue(std::integral_constant<int, -1>{});
}

// ------------------------------------------------------------------------------------------------
// Generic fp32 kernel: v_mfma_f32_16x16x4_f32 == an fmaf chain in k order (exact f32).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_generic32(GemmArgs p) {
    constexpr int BM = 64, BN = 64, BK = 16, PITCH = BK + 1;
    __shared__ float As[BM][PITCH];
    __shared__ float Bs[BN][PITCH];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int ib = blockIdx.z;
    const int m0 = (blockIdx.x / p.tiles_n) * BM, n0 = (blockIdx.x % p.tiles_n) * BN;
    const float *A = (const float *)p.a + (long)ib * p.a_bs;
    const float *B = (const float *)p.b + (long)ib * p.b_bs;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool a_kc = (p.a_cs == 1), b_kc = (p.b_rs == 1);
    for (int k0 = 0; k0 < p.k; k0 += BK) {
#pragma unroll
        for (int j = 0; j < (BM * BK) / 256; ++j) {
            const int idx = t + j * 256;
            int i, kk;
            if (a_kc) { kk = idx % BK; i = idx / BK; } else { i = idx % BM; kk = idx / BM; }
            const int gi = m0 + i, gk = k0 + kk;
            float v = 0.f;
            if (gi < p.m && gk < p.k)
                v = A[(long)gi * p.a_rs + (long)gk * p.a_cs];
            As[i][kk] = v;
        }
#pragma unroll
        for (int j = 0; j < (BN * BK) / 256; ++j) {
            const int idx = t + j * 256;
            int jn, kk;
            if (b_kc) { kk = idx % BK; jn = idx / BK; } else { jn = idx % BN; kk = idx / BN; }
            const int gj = n0 + jn, gk = k0 + kk;
            float v = 0.f;
            if (gj < p.n && gk < p.k)
                v = B[(long)gk * p.b_rs + (long)gj * p.b_cs];
            Bs[jn][kk] = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[i] = As[wm * 32 + i * 16 + (lane & 15)][kk * 4 + (lane >> 4)];
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bf[j] = Bs[wn * 32 + j * 16 + (lane & 15)][kk * 4 + (lane >> 4)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    float *C = (float *)p.c + (long)ib * p.c_bs;
    const float *bias = (const float *)p.bias;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
                const int col = n0 + wn * 32 + j * 16 + (lane & 15);
                if (row < p.m && col < p.n) {
                    float v = acc[i][j][r];
                    if (bias)
                        v += bias[(long)ib * p.bias_b + (long)row * p.bias_m + (long)col * p.bias_n];
                    C[c_off(p, row, col)] = apply_act(v, p.act);
                }
            }
}

// ------------------------------------------------------------------------------------------------
// Fast 16-bit kernel, 128x128x64, LDS-DMA double buffer.
// ------------------------------------------------------------------------------------------------

template <typename Tr, bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(256) void gemm_fast128(GemmArgs p) {
    using namespace f128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto a_tile = [&](int buf) -> char * { return smem + buf * TILE_BYTES; };
    auto b_tile = [&](int buf) -> char * { return smem + (2 + buf) * TILE_BYTES; };

    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = w >> 1, wn = w & 1;

    // workgroup -> (batch, tile_m, tile_n): XCD-aware remap, then grouped raster (8 tile-rows).
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
    const long lda = A_KMAJOR ? p.a_rs : p.a_cs; // leading dimension of the stored matrix
    const long ldb = B_KMAJOR ? p.b_cs : p.b_rs;

    // chunks of the last K-tile that start at k >= p.k come from a block of zeros (k % 8 == 0 but k % 64 != 0)
    const unsigned short *Z = (const unsigned short *)p.zeros;
    auto stage = [&](int buf, int k0) {
        if constexpr (A_KMAJOR)
            stage_kmajor(A, lda, m0, p.m, k0, a_tile(buf), w, lane, p.k, Z);
        else
            stage_mnmajor(A, lda, m0, p.m, k0, a_tile(buf), w, lane, p.k, Z);
        if constexpr (B_KMAJOR)
            stage_kmajor(B, ldb, n0, p.n, k0, b_tile(buf), w, lane, p.k, Z);
        else
            stage_mnmajor(B, ldb, n0, p.n, k0, b_tile(buf), w, lane, p.k, Z);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.k + BK - 1) / BK;
    // One K-tile step with a COMPILE-TIME buffer index: hipcc only keeps the LDS-DMA of the next
    // tile in flight across the ds_reads of the current one when it can prove the two LDS ranges
    // distinct; a runtime (kt & 1) index makes it drain vmcnt(0) before the first ds_read.
    auto step = [&](auto bufc, int kt) {
        constexpr int buf = decltype(bufc)::value;
        // tile kt has landed (own DMA waited for, then barrier => everybody's), and every wave has
        // finished reading the other buffer (its compute of tile kt-1 precedes this barrier).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk)
            stage(buf ^ 1, (kt + 1) * BK);
        const char *at = a_tile(buf), *bt = b_tile(buf);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            s16x8_t af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i] = A_KMAJOR ? frag_kmajor(at, wm * 64 + i * 16, ks, lane)
                                 : frag_mnmajor(at, wm * 64 + i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bf[j] = B_KMAJOR ? frag_kmajor(bt, wn * 64 + j * 16, ks, lane)
                                 : frag_mnmajor(bt, wn * 64 + j * 16, ks, lane);
            // swapped operands: D[i][j] = sum_k Bop[k][i] Aop[j][k] = C[m = j][n = i], so a lane
            // ends up with 4 consecutive n of one m row (vector stores in the epilogue).
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = Tr::mfma(bf[j], af[i], acc[i][j]);
        }
    };
    stage(0, 0);
    for (int kt = 0; kt < nk; kt += 2) {
        step(std::integral_constant<int, 0>{}, kt);
        if (kt + 1 < nk)
            step(std::integral_constant<int, 1>{}, kt + 1);
    }

    // one copy of the epilogue per activation class (0 none, 1 relu, 5 A-S Gelu, -1 = p.act at run time): see gemm256p_kernel.h
    auto epilogue = [&](auto actc) __attribute__((always_inline)) {
        constexpr int ACT = decltype(actc)::value;
        auto act1 = [&](float v) {
            if constexpr (ACT == 1) return v > 0.f ? v : 0.f;
            else if constexpr (ACT == 5) return gelu_poly(v);
            else if constexpr (ACT < 0) return apply_act(v, p.act);
            else return v;
        };
        unsigned short *C = (unsigned short *)p.c + (long)ib * p.c_bs;
        const unsigned short *bias = (const unsigned short *)p.bias;
        const bool interior = (m0 + BM <= p.m) && (n0 + BN <= p.n) && (p.n % 4 == 0);
        if (interior && (p.n % 8 == 0) && ((((uintptr_t)p.c) & 15) == 0)) {
            // 16-byte stores by swapping half tiles between lane groups g4 / g4^1 (same exchange as gemm256.hip)
            const int l15 = lane & 15, g4 = lane >> 4;
            const bool odd = g4 & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = m0 + wm * 64 + i * 16 + l15;
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    unsigned pk[2][2];
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        const int col = n0 + wn * 64 + (jp * 2 + t2) * 16 + g4 * 4;
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
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            v[r] = act1(v[r]);
                        pk[t2][0] = (unsigned)Tr::from_f32(v[0]) | ((unsigned)Tr::from_f32(v[1]) << 16);
                        pk[t2][1] = (unsigned)Tr::from_f32(v[2]) | ((unsigned)Tr::from_f32(v[3]) << 16);
                    }
                    const unsigned s0 = odd ? pk[0][0] : pk[1][0], s1 = odd ? pk[0][1] : pk[1][1];
                    const unsigned r0 = (unsigned)__shfl_xor((int)s0, 16), r1 = (unsigned)__shfl_xor((int)s1, 16);
                    u32x4_t o;
                    if (odd) { o[0] = r0; o[1] = r1; o[2] = pk[1][0]; o[3] = pk[1][1]; }
                    else { o[0] = pk[0][0]; o[1] = pk[0][1]; o[2] = r0; o[3] = r1; }
                    const int col = n0 + wn * 64 + (jp * 2 + (odd ? 1 : 0)) * 16 + (g4 & ~1) * 4;
                    *(u32x4_t *)(C + c_off(p, row, col)) = o;
                }
            }
        } else if (interior) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = m0 + wm * 64 + i * 16 + (lane & 15);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
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
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        v[r] = act1(v[r]);
                    u32x2_t pk;
                    pk[0] = (unsigned)Tr::from_f32(v[0]) | ((unsigned)Tr::from_f32(v[1]) << 16);
                    pk[1] = (unsigned)Tr::from_f32(v[2]) | ((unsigned)Tr::from_f32(v[3]) << 16);
                    *(u32x2_t *)(C + c_off(p, row, col)) = pk;
                }
            }
        } else {
            for (int i = 0; i < 4; ++i) {
                const int row = m0 + wm * 64 + i * 16 + (lane & 15);
                for (int j = 0; j < 4; ++j) {
                    const int col = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
                    for (int r = 0; r < 4; ++r) {
                        // select the register with a static index (runtime-indexed vectors go to scratch)
                        float v = 0.f;
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                                for (int rr = 0; rr < 4; ++rr)
                                    if (ii == i && jj == j && rr == r)
                                        v = acc[ii][jj][rr];
                        if (row < p.m && col + r < p.n) {
                            if (bias)
                                v += Tr::to_f32(bias[(long)ib * p.bias_b + (long)row * p.bias_m + (long)(col + r) * p.bias_n]);
                            C[c_off(p, row, col + r)] = Tr::from_f32(act1(v));
                        }
                    }
                }
            }
        }
    };
    if (p.act == 0) epilogue(std::integral_constant<int, 0>{});
    else if (p.act == 1) epilogue(std::integral_constant<int, 1>{});
    else if (p.act == 5) epilogue(std::integral_constant<int, 5>{});
    else epilogue(std::integral_constant<int, -1>{});
}

// implemented in gemm256.hip
int launch_gemm256(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, bool a_kmajor, bool b_kmajor);
bool gemm256_supported(const GemmArgs &p, bool a_kmajor, bool b_kmajor);
namespace g256p { // persistent multi-tile kernels, tile 256 x 64 NT (gemm256p_kernel.h)
int launch_gemm256p_nt4(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, bool akm, bool bkm);
int launch_gemm256p_nt3(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, bool akm, bool bkm);
int launch_gemm256p_nt2(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, bool akm, bool bkm);
int launch_gemm256p_trace(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, int nt, unsigned long long *trace);
} // namespace g256p
int launch_gemm256_splitk(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, bool a_kmajor, bool b_kmajor, int splits);
int launch_gemm256_f32out(infiniRocmRuntime_t rt, int dtype16, const GemmArgs &p, bool a_kmajor, bool b_kmajor, int splits, float *planes);
namespace g128w { // four waves x 128 x 128 wave tiles, four-stage ring (gemm128w.hip): plain GEMMs on whole 256^2 tiles, K % 128 == 0
bool supported(const GemmArgs &p);
int launch_gemm128w(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, bool akm, bool bkm);
} // namespace g128w
// implemented in gemm32.hip: the fp32 128^2 LDS-DMA tile kernel (v_mfma_f32_32x32x2_f32)
bool fast32_supported(const GemmArgs &p, bool a_kmajor, bool b_kmajor);
int launch_fast32(infiniRocmRuntime_t rt, GemmArgs p, bool b_kmajor, int small_tiles);

template <typename Tr> static int launch_fast128(infiniRocmRuntime_t rt, GemmArgs p, bool akm, bool bkm) {
    p.tiles_m = (int)ceil_div(p.m, f128::BM);
    p.tiles_n = (int)ceil_div(p.n, f128::BN);
    const unsigned grid = (unsigned)p.tiles_m * p.tiles_n * p.batch;
    const size_t lds = 4 * f128::TILE_BYTES;
#define IROCM_F128(AK, BK_)                                                                        \
    do {                                                                                           \
        auto kern = gemm_fast128<Tr, AK, BK_>;                                                     \
        IROCM_LDS_ATTR(kern, (int)lds, rt);                                                        \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, rt->stream, p);                       \
    } while (0)
    if (akm && bkm) IROCM_F128(true, true);
    else if (akm && !bkm) IROCM_F128(true, false);
    else if (!akm && bkm) IROCM_F128(false, true);
    else IROCM_F128(false, false);
#undef IROCM_F128
    IROCM_LAUNCH_CHECK("gemm_fast128");
    return INFINI_ROCM_OK;
}

static bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

static bool fast128_supported(const GemmArgs &p, bool akm, bool bkm) {
    if (p.k % 8 != 0 || p.k < 8) // 16-byte K runs; a K tail inside the last 64-wide tile is zero-filled
        return false;
    if (!aligned16(p.a) || !aligned16(p.b) || (p.a_bs % 8) || (p.b_bs % 8))
        return false;
    if (!akm && (p.m % 8 != 0 || p.m < 8))
        return false;
    if (!bkm && (p.n % 8 != 0 || p.n < 8))
        return false;
    if (!aligned16(p.c) && (p.n % 4 == 0))
        return false;
    return true;
}

// gemm128w.hip keeps a lane's byte offset from the tile corner — the rows of its pieces PLUS the whole k advance of a tile — in 32 bits
static bool w128_offsets_fit(const GemmArgs &p, bool akm, bool bkm) {
    const long lda = akm ? p.a_rs : p.a_cs, ldb = bkm ? p.b_cs : p.b_rs;
    const long a_max = akm ? (255 * lda + p.k + 64) * 2 : ((long)(p.k + 32) * lda + 256) * 2;
    const long b_max = bkm ? (255 * ldb + p.k + 64) * 2 : ((long)(p.k + 32) * ldb + 256) * 2;
    return a_max < (1ll << 32) && b_max < (1ll << 32);
}

static const char *kVariantNames[] = {"generic64", "fast128_glds", "tile256", "tile256_splitk", "persist256", "persist192",
                                      "persist128", "fast32", "wave128"};
constexpr int kNumVariants = 9; // 1-6 and 8 serve f16 / bf16, 7 serves f32, 0 everything

// Cost model behind the heuristic (microseconds; fitted to tools/gemm_shapes.py on MI355X, bf16 / f16, N(0,1) data).
// A workgroup of the persistent kernel walks its tiles: a K-tile of a 256 x 64 NT tile costs kKt[NT]; every tile pays its
// tile boundary (both wave rows' epilogues side by side + the pipeline restart; gemm256p_kernel.h); launch + first prologue
// ~3 us once. Re-fitted after the epilogues were de-serialised (round 2: ~12.4 k cycles per boundary) and again after their
// stores went quad-contiguous (round 3: ~7.9 k cycles; profiles/r03_gemm_shapes_bf16.txt).
static const double kKt[5] = {0, 0, 0.91, 1.10, 1.40};
static const double kStoreTail[5] = {0, 0, 4.2, 5.2, 5.5};
static double persist_cost(long m, long n, long k, long batch, int nt, int cus) {
    const long tiles = ceil_div(m, 256) * ceil_div(n, 64 * nt) * batch;
    const long full = tiles / cus;
    const double frac = (double)(tiles - full * cus) / cus;
    // a partial last round still costs most of a tile time (every workgroup's tile takes what it takes; only the shared
    // L2 / HBM / power budget is lighter): 0.55 + 0.5 frac of a full round fits the sweep from frac = 0.25 to 0.8
    const double waves = (double)full + (frac > 0 ? (0.55 + 0.5 * frac < 1.0 ? 0.55 + 0.5 * frac : 1.0) : 0.0);
    return waves * ((double)(k / 64) * kKt[nt] + kStoreTail[nt]) + 3.0;
}
// split-K: `splits` workgroups per 256^2 tile write fp32 partial planes, one reduce pass adds them
static double splitk_cost(long m, long n, long k, long batch, int splits) {
    return 3.0 + (double)(k / 64) / splits * kKt[4] + 12.0 + (double)batch * m * n * (4.0 * splits + 2.0) / 5.0e6;
}

// tile width (NT = 4 / 3 / 2 -> 256 / 192 / 128 columns) the cost model prefers for an m x n x k problem on the persistent
// kernels; max_nt caps it (the conv mode's residual copy exists up to NT = 3)
int persist_pick_nt(long m, long n, long k, int cus, int max_nt) {
    int best_nt = max_nt < 4 ? max_nt : 4;
    double best = 1e30;
    for (int nt = best_nt; nt >= 2; --nt) {
        const double c = persist_cost(m, n, k, 1, nt, cus);
        if (c < best * 0.97) {
            best = c;
            best_nt = nt;
        }
    }
    return best_nt;
}

} // namespace irocm

using namespace irocm;

extern "C" {

int infini_rocm_matmul_num_variants(void) { return kNumVariants; }

const char *infini_rocm_matmul_variant_name(int v) {
    return (v >= 0 && v < kNumVariants) ? kVariantNames[v] : "invalid";
}

int infini_rocm_matmul_last_variant(infiniRocmRuntime_t rt, int *variant) {
    IROCM_CHECK_ARG(rt && variant, "NULL argument");
    *variant = rt->last_matmul_variant;
    return INFINI_ROCM_OK;
}

// MatmulObj::getComputeType() (reference: matmul.cc:51-64 — "tf32" / "fp16" / "bf16" select cuBLAS compute types whose PRODUCTS
// take reduced-precision inputs while sums and outputs stay fp32). 0 "default" / "tf32": exact fp32 products (gfx950 has no
// xf32 MFMA; more accurate than asked). 1 "bf16", 2 "fp16": an fp32 MatMul converts A and B once into the workspace and runs the
// 16-bit MFMA kernel with fp32 accumulation and fp32 output — the reference's opt-in ~10x over exact fp32. Sticky per runtime
// (the plugin sets it around the one launch); ignored for 16-bit operands.
int infini_rocm_matmul_set_compute_type(infiniRocmRuntime_t rt, int compute_type) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(compute_type >= 0 && compute_type <= 2, "compute type %d: 0 default / tf32, 1 bf16, 2 fp16", compute_type);
    rt->matmul_compute_type = compute_type;
    return INFINI_ROCM_OK;
}

int infini_rocm_matmul_set_variant(infiniRocmRuntime_t rt, int variant) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(variant >= -1 && variant < kNumVariants, "variant %d out of range", variant);
    rt->matmul_variant = variant;
    return INFINI_ROCM_OK;
}

// split-K (the only MatMul path that takes the runtime workspace: fp32 partial planes) needs few enough 256^2 tiles
static bool splitk_possible(infiniRocmRuntime_t rt, int64_t m, int64_t n, int64_t batch) {
    const long tiles256 = ceil_div(m, 256) * ceil_div(n, 256) * batch;
    return tiles256 * 2 <= rt->num_cu + rt->num_cu / 4;
}

int infini_rocm_matmul_may_use_workspace(infiniRocmRuntime_t rt, int64_t batch, int64_t m, int64_t n, int *may) {
    IROCM_CHECK_ARG(rt && may, "NULL argument");
    *may = (rt->matmul_variant == 3 || splitk_possible(rt, m, n, batch)) ? 1 : 0;
    return INFINI_ROCM_OK;
}

int infini_rocm_matmul(infiniRocmRuntime_t rt, int dtype, const void *a, const void *b,
                       const void *bias, void *c, int64_t batch, int64_t m, int64_t n, int64_t k,
                       int trans_a, int trans_b, int64_t stride_a, int64_t stride_b,
                       int64_t bias_stride_b, int64_t bias_stride_m, int64_t bias_stride_n,
                       int act) {
    return infini_rocm_matmul_headsplit(rt, dtype, a, b, bias, c, batch, m, n, k, trans_a, trans_b, stride_a, stride_b,
                                        bias_stride_b, bias_stride_m, bias_stride_n, act, 0, 0);
}

int infini_rocm_matmul_headsplit(infiniRocmRuntime_t rt, int dtype, const void *a, const void *b,
                                 const void *bias, void *c, int64_t batch, int64_t m, int64_t n, int64_t k,
                                 int trans_a, int trans_b, int64_t stride_a, int64_t stride_b,
                                 int64_t bias_stride_b, int64_t bias_stride_m, int64_t bias_stride_n,
                                 int act, int64_t seq, int64_t head_dim) {
    return infini_rocm_matmul_grouped(rt, dtype, a, b, bias, c, batch, m, n, k, trans_a, trans_b, stride_a, stride_b, 0,
                                      bias_stride_b, bias_stride_m, bias_stride_n, act, seq, head_dim);
}

int infini_rocm_matmul_grouped(infiniRocmRuntime_t rt, int dtype, const void *a, const void *b,
                               const void *bias, void *c, int64_t batch, int64_t m, int64_t n, int64_t k,
                               int trans_a, int trans_b, int64_t stride_a, int64_t stride_b, int64_t stride_c,
                               int64_t bias_stride_b, int64_t bias_stride_m, int64_t bias_stride_n,
                               int act, int64_t seq, int64_t head_dim) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(stride_c == 0 || stride_c >= m * n || stride_c <= -(m * n) || batch <= 1,
                    "matmul: output blocks of %lld elements overlap at a batch stride of %lld", (long long)(m * n), (long long)stride_c);
    IROCM_CHECK_ARG(stride_c % 8 == 0, "matmul: the output batch stride must be a multiple of 8 elements (16-byte stores)");
    IROCM_CHECK_ARG((seq == 0) == (head_dim == 0), "matmul: seq and head_dim go together");
    if (head_dim) {
        IROCM_CHECK_ARG(seq > 0 && head_dim > 0 && head_dim % 8 == 0 && m % seq == 0 && n % head_dim == 0 &&
                            seq < (1ll << 31) && head_dim < (1ll << 31),
                        "matmul: head split [m/%lld][n/%lld][%lld][%lld] does not tile m = %lld, n = %lld (head_dim %% 8 == 0)",
                        (long long)seq, (long long)head_dim, (long long)seq, (long long)head_dim, (long long)m, (long long)n);
    }
    IROCM_CHECK_ARG(dtype == INFINI_DT_F32 || dtype == INFINI_DT_F16 || dtype == INFINI_DT_BF16,
                    "matmul: unsupported dtype %s", dtype_name(dtype));
    IROCM_CHECK_ARG(batch >= 0 && m >= 0 && n >= 0 && k >= 0, "matmul: negative dimension");
    IROCM_CHECK_ARG(batch < 65536 && m < (1ll << 31) && n < (1ll << 31) && k < (1ll << 31),
                    "matmul: dimension too large");
    IROCM_CHECK_ARG(act >= 0 && act <= 5, "matmul: bad act %d", act);
    if (batch == 0 || m == 0 || n == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(a && b && c, "matmul: NULL operand");

    GemmArgs p;
    p.a = a; p.b = b; p.bias = bias; p.c = c;
    p.m = (int)m; p.n = (int)n; p.k = (int)k; p.batch = (int)batch;
    p.a_rs = trans_a ? 1 : k; p.a_cs = trans_a ? m : 1; p.a_bs = stride_a;
    p.b_rs = trans_b ? 1 : n; p.b_cs = trans_b ? k : 1; p.b_bs = stride_b;
    p.bias_b = bias_stride_b; p.bias_m = bias_stride_m; p.bias_n = bias_stride_n;
    p.c_bs = stride_c ? stride_c : m * n;
    p.act = act;
    p.tiles_m = p.tiles_n = 0;
    p.splitk = 1;
    p.partial = nullptr;
    p.zeros = rt->zeros;
    p.epi16 = 1; // 16-byte epilogue stores
    p.hs_s = (int)seq;
    p.hs_d = (int)head_dim;
    const bool akm = !trans_a, bkm = trans_b != 0;

    // (batch strides: the casts below copy (stride ? batch : 1) CONTIGUOUS blocks of m * k (n * k) elements, so the path is taken only
    // for operands that ARE such blocks — stride 0 (shared) or exactly one block; any other stride keeps the exact kernels.
    // Round-4 advisor: with another stride the cast read the wrong rows and the kernel indexed the 16-bit copy past its end.)
    const bool ct_strides = (stride_a == 0 || stride_a == m * k || batch == 1) && (stride_b == 0 || stride_b == n * k || batch == 1);
    if (dtype == INFINI_DT_F32 && rt->matmul_compute_type != 0 && head_dim == 0 && stride_c == 0 && ct_strides) {
        // reduced-precision products on request: 16-bit copies of A and B in the workspace, the 256^2 split-K kernel (raw fp32
        // slice sums), fp32 output. Shapes it cannot serve (K % 64, alignment) keep the exact kernels: never LESS accurate than asked.
        const int dt16 = rt->matmul_compute_type == 1 ? INFINI_DT_BF16 : INFINI_DT_F16;
        const bool a_shared = stride_a == 0 || batch == 1, b_shared = stride_b == 0 || batch == 1;
        const int64_t na = (a_shared ? 1 : batch) * m * k, nb = (b_shared ? 1 : batch) * n * k;
        GemmArgs q = p;
        q.a_bs = a_shared ? 0 : m * k;
        q.b_bs = b_shared ? 0 : n * k;
        const size_t a_bytes = ((size_t)na * 2 + 255) & ~(size_t)255, b_bytes = ((size_t)nb * 2 + 255) & ~(size_t)255;
        // the support test reads alignment and shape only: a 256-byte-aligned stand-in for the workspace pointers decides it BEFORE
        // the workspace is grown (an unsupported shape must not cost an allocation)
        q.a = (const void *)(uintptr_t)256;
        q.b = (const void *)(uintptr_t)(256 + a_bytes);
        if (gemm256_supported(q, akm, bkm) && (((uintptr_t)c) & 15) == 0) {
            const long tiles = ceil_div(m, 256) * ceil_div(n, 256) * batch;
            int splits = (int)std::max<long>(1, rt->num_cu / tiles);
            splits = std::min(splits, std::max(1, (int)(k / 512)));
            splits = std::min(splits, 16);
            const size_t plane_bytes = (size_t)splits * batch * m * n * sizeof(float);
            char *ws = nullptr;
            int st = infini_rocm_workspace(rt, a_bytes + b_bytes + plane_bytes, (void **)&ws);
            if (st != INFINI_ROCM_OK)
                return st;
            q.a = ws;
            q.b = ws + a_bytes;
            st = infini_rocm_cast(rt, INFINI_DT_F32, dt16, a, ws, na);
            if (st == INFINI_ROCM_OK)
                st = infini_rocm_cast(rt, INFINI_DT_F32, dt16, b, ws + a_bytes, nb);
            if (st != INFINI_ROCM_OK)
                return st;
            rt->last_matmul_variant = 3;
            return launch_gemm256_f32out(rt, dt16, q, akm, bkm, splits, (float *)(ws + a_bytes + b_bytes));
        }
    }
    int variant = rt->matmul_variant;
    if (dtype == INFINI_DT_F32) {
        // fp32: the LDS-DMA tile kernel (gemm32.hip; 128^2 or 64^2 tiles) when it can serve the operands and the problem has
        // at least 16 tiles of 64^2 (or it is forced); the generic register-staged 64^2 kernel otherwise
        const long tiles64 = ceil_div(m, 64) * ceil_div(n, 64) * batch;
        const bool want = variant == 7 || (variant < 0 && tiles64 >= 16 && k >= 64);
        variant = (want && fast32_supported(p, akm, bkm)) ? 7 : 0;
    } else if (variant == 7) {
        variant = -1;
    } else if (variant == 8 && !(g128w::supported(p) && w128_offsets_fit(p, akm, bkm))) {
        variant = -1;
    }
    // split-K factor for the 256^2 kernel: fill the CUs when the tiles alone cannot and K is long enough that every
    // slice still runs >= 8 K-tiles (the fp32 partial planes cost 8 bytes per output element and slice)
    const long tiles256 = ceil_div(m, 256) * ceil_div(n, 256) * batch;
    int splits = 1;
    if (gemm256_supported(p, akm, bkm) && splitk_possible(rt, m, n, batch)) {
        splits = (int)(rt->num_cu / tiles256);
        const int max_by_k = (int)(k / (8 * 64));
        if (splits > max_by_k) splits = max_by_k;
        if (splits > 16) splits = 16;
    }
    if (variant < 0) {
        // heuristic: the cheapest of {persistent 256 / 192 / 128-wide tiles, split-K} by the cost model when the 256-row
        // kernels can serve the problem and it has at least ~half a tile per CU; otherwise 128^2 tiles; otherwise generic
        const bool ok256 = gemm256_supported(p, akm, bkm);
        double best = 1e30;
        if (ok256) {
            for (int nt = 4; nt >= 2; --nt) {
                const long tiles = ceil_div(m, 256) * ceil_div(n, 64 * nt) * batch;
                if (tiles * 2 < rt->num_cu)
                    continue;
                const double c = persist_cost(m, n, k, batch, nt, rt->num_cu);
                if (c < best * 0.97) { // prefer the wider tile unless a narrower one is clearly cheaper
                    best = c;
                    variant = 4 + (4 - nt);
                }
            }
            if (splits >= 2 && splitk_cost(m, n, k, batch, splits) < best * 0.97)
                variant = 3;
        }
        // the four-wave kernel (gemm128w.hip) where it measured ahead of persist256 (profiles/r06_gemm_wave128_ab.txt: + 2-8 %): plain
        // single-batch GEMMs of one or two rounds of whole 256^2 tiles with a long K, any layout but NT (both operands K-major: - 3.5 %)
        if (variant == 4 && batch == 1 && k >= 2048 && !(akm && bkm) && tiles256 >= rt->num_cu && tiles256 <= 2l * rt->num_cu &&
            g128w::supported(p) && w128_offsets_fit(p, akm, bkm))
            variant = 8;
        if (variant < 0)
            variant = fast128_supported(p, akm, bkm) ? 1 : 0;
    } else if (dtype == INFINI_DT_F32) {
        // 0 or 7, decided above: the 16-bit kernels below never see fp32 operands
    } else if (variant >= 2 && !gemm256_supported(p, akm, bkm)) {
        variant = fast128_supported(p, akm, bkm) ? 1 : 0;
    } else if (variant == 1 && !fast128_supported(p, akm, bkm)) {
        variant = 0;
    }

    // sigmoid / tanh / erff-Gelu epilogues and biases other than one row vector live in the one-shot kernel (gemm256p_kernel.h)
    if (variant >= 4 && variant <= 6 && (!(act == 0 || act == 1 || act == 5) || (p.bias && !(p.bias_m == 0 && p.bias_n == 1))))
        variant = 2;
    rt->last_matmul_variant = variant;
    if (variant == 7) // 128^2 tiles when they give at least ~half a tile per CU, 64^2 tiles otherwise (512^3: 64 tiles)
        return launch_fast32(rt, p, bkm, ceil_div(m, 128) * ceil_div(n, 128) * batch * 2 < rt->num_cu ? 1 : 0);
    if (variant == 8)
        return g128w::launch_gemm128w(rt, dtype, p, akm, bkm);
    if (variant == 4)
        return g256p::launch_gemm256p_nt4(rt, dtype, p, akm, bkm);
    if (variant == 5)
        return g256p::launch_gemm256p_nt3(rt, dtype, p, akm, bkm);
    if (variant == 6)
        return g256p::launch_gemm256p_nt2(rt, dtype, p, akm, bkm);
    if (variant == 3)
        return launch_gemm256_splitk(rt, dtype, p, akm, bkm, splits < 2 ? 2 : splits);
    if (variant == 2)
        return launch_gemm256(rt, dtype, p, akm, bkm);
    if (variant == 1)
        return dtype == INFINI_DT_BF16 ? launch_fast128<Bf16Traits>(rt, p, akm, bkm)
                                       : launch_fast128<F16Traits>(rt, p, akm, bkm);
    p.tiles_m = (int)ceil_div(m, 64);
    p.tiles_n = (int)ceil_div(n, 64);
    IROCM_CHECK_ARG((int64_t)p.tiles_m * p.tiles_n < (1ll << 31), "matmul: too many tiles");
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n), 1, (unsigned)batch);
    if (dtype == INFINI_DT_F32)
        hipLaunchKernelGGL(gemm_generic32, grid, dim3(256), 0, rt->stream, p);
    else if (dtype == INFINI_DT_BF16)
        hipLaunchKernelGGL(gemm_generic16<Bf16Traits>, grid, dim3(256), 0, rt->stream, p);
    else
        hipLaunchKernelGGL(gemm_generic16<F16Traits>, grid, dim3(256), 0, rt->stream, p);
    IROCM_LAUNCH_CHECK("gemm_generic");
    return INFINI_ROCM_OK;
}

// Diagnostics: one launch of the persistent GEMM (bf16, NN, no bias) with every wave stamping s_memtime at its phase
// boundaries; trace receives [min(tiles, CUs)][8][128] stamps (0 = unused slot): slot 0 kernel entry, then per K-tile
// {L1 start, L2 start}, per tile {epilogue start, epilogue end}, last = after the final store drain.
int infini_rocm_probe_gemm_timeline(infiniRocmRuntime_t rt, const void *a, const void *b, void *c, int64_t m, int64_t n,
                                    int64_t k, int tile_cols, void *trace) {
    IROCM_CHECK_ARG(rt && a && b && c && trace, "probe: NULL argument");
    IROCM_CHECK_ARG(tile_cols == 256 || tile_cols == 192, "probe: tile_cols must be 256 or 192");
    GemmArgs p;
    memset(&p, 0, sizeof(p));
    p.a = a; p.b = b; p.c = c;
    p.m = (int)m; p.n = (int)n; p.k = (int)k; p.batch = 1;
    p.a_rs = k; p.a_cs = 1; p.b_rs = n; p.b_cs = 1;
    p.splitk = 1; p.epi16 = 1; p.zeros = rt->zeros;
    IROCM_CHECK_ARG(gemm256_supported(p, true, false), "probe: shape not served by the 256-row kernels");
    return g256p::launch_gemm256p_trace(rt, INFINI_DT_BF16, p, tile_cols / 64, (unsigned long long *)trace);
}

} // extern "C"
