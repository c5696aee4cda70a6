// Diagnostics: the MFMA-only ceiling of this chip under ITS power budget.
//
// MI355X clocks to its power limit (MI355X_MICROARCH.md, "DVFS give-back"): the nominal dense bf16 peak (256 CUs x
// 4096 FLOP/clk x 2.4 GHz = 2.5 PFLOP/s) is not what a matrix-pipe-saturating kernel on random data can reach. This
// kernel issues the SAME MFMA stream as the headline GEMM (8 waves per workgroup = 2 per SIMD, 8 x 4 accumulator tiles
// of v_mfma_f32_16x16x32 per wave, 12 distinct operand fragments of the caller's data, one workgroup per CU) with NO
// memory or LDS instruction in the loop. Its TFLOP/s is the measured denominator for "how far is the GEMM from what the
// matrix pipes can do at the clock the chip sustains" (bench.py: roofline.attainable_peak).
#include "gemm_common.h"

namespace irocm {

template <typename Tr>
__global__ __launch_bounds__(512, 2) void mfma_ceiling_kernel(const unsigned short *__restrict__ data, float *__restrict__ sink,
                                                              int iters) {
    const int t = threadIdx.x;
    // 12 operand fragments per lane (8 A, 4 B), 16 bytes each, distinct per lane and per workgroup slot
    const s16x8_t *src = (const s16x8_t *)data + ((size_t)(blockIdx.x & 15) * 512 + t) * 12;
    s16x8_t a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = src[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = src[8 + j];
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = Tr::mfma(b[j], a[(i + ks) & 7], acc[i][j]);
        // keep the loop a loop (no cross-iteration folding), no memory traffic
        asm volatile("" : "+v"(a[0]), "+v"(b[0]));
    }
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            s += acc[i][j];
    if (s[0] + s[1] + s[2] + s[3] == 123.456f) // practically never: keeps the accumulators live
        sink[blockIdx.x * 512 + t] = s[0];
}

// The same FLOPs per wave and iteration on v_mfma_f32_32x32x16 (4 x 2 accumulator tiles of 32 x 32, four k-steps of 16): round-5
// experiment (verdict item 3c) — does the 32 x 32 shape, whose microbenchmark ceiling is 15 % above the 16 x 16 one, sustain more
// under the chip's power budget on the GEMM's operand data?
template <typename Tr>
__global__ __launch_bounds__(512, 2) void mfma_ceiling32_kernel(const unsigned short *__restrict__ data, float *__restrict__ sink, int iters) {
    const int t = threadIdx.x;
    const s16x8_t *src = (const s16x8_t *)data + ((size_t)(blockIdx.x & 15) * 512 + t) * 12;
    s16x8_t a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = src[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = src[8 + j];
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (Tr::kDType == INFINI_DT_BF16)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, b[(j + ks) & 3]),
                                                                            __builtin_bit_cast(bf16x8_t, a[(i + 2 * ks) & 7]), acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, b[(j + ks) & 3]),
                                                                           __builtin_bit_cast(f16x8_t, a[(i + 2 * ks) & 7]), acc[i][j], 0, 0, 0);
                }
        asm volatile("" : "+v"(a[0]), "+v"(b[0]));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                s += acc[i][j][e];
    if (s == 123.456f)
        sink[blockIdx.x * 512 + t] = s;
}

} // namespace irocm

using namespace irocm;

// As infini_rocm_probe_mfma_ceiling on v_mfma_f32_32x32x16: iters * 32 MFMAs per wave, the same FLOP count per iteration.
extern "C" int infini_rocm_probe_mfma_ceiling32(infiniRocmRuntime_t rt, int dtype, const void *data, void *sink, int iters, double *flop) {
    IROCM_CHECK_ARG(rt && data && sink && iters > 0, "probe: bad argument");
    IROCM_CHECK_ARG(dtype == INFINI_DT_BF16 || dtype == INFINI_DT_F16, "probe: bf16 / f16 only");
    const unsigned grid = (unsigned)rt->num_cu;
    if (dtype == INFINI_DT_BF16)
        hipLaunchKernelGGL(mfma_ceiling32_kernel<Bf16Traits>, dim3(grid), dim3(512), 0, rt->stream, (const unsigned short *)data, (float *)sink, iters);
    else
        hipLaunchKernelGGL(mfma_ceiling32_kernel<F16Traits>, dim3(grid), dim3(512), 0, rt->stream, (const unsigned short *)data, (float *)sink, iters);
    IROCM_LAUNCH_CHECK("mfma_ceiling32");
    if (flop)
        *flop = (double)grid * 8.0 * (double)iters * 32.0 * (2.0 * 32 * 32 * 16);
    return INFINI_ROCM_OK;
}

// Round 6 (verdict item 1, the second design it names): the GEMM's MFMA stream with the A operand streamed L2 -> VGPR, never through LDS.
// An UPPER bound for that design: B fragments stay in registers for the whole launch (no B traffic at all), no barrier, no LDS; per
// K-tile of 64 a wave loads exactly what the 256 x 256 x 64 kernel's wave would need of A — 128 rows x 64 k = 16 fragment-shaped
// global_load_dwordx4 (16 rows x 64 bytes per instruction) — from a [rows][k] panel that stays L2-resident (every workgroup of an XCD
// walks the same 2 MB), register double-buffered by halves (the 32 MFMAs of one 64-row half run while the other half's 8 loads fly).
template <typename Tr>
__global__ __launch_bounds__(512, 2) void mfma_a_from_l2_kernel(const unsigned short *__restrict__ a, const unsigned short *__restrict__ bdata,
                                                                float *__restrict__ sink, int k, int iters) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wr = w >> 2;
    s16x8_t b[4][2];
    const s16x8_t *bsrc = (const s16x8_t *)bdata + ((size_t)(blockIdx.x & 15) * 512 + t) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            b[j][ks] = bsrc[j * 2 + ks];
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this lane's fragment of row block i, k-step ks of K-tile kt: row (blockIdx.x % 8) * 256 + wr * 128 + i * 16 + l15, k = kt * 64 + ks * 32 + g4 * 8
    const unsigned short *row0 = a + ((size_t)((blockIdx.x & 7) * 256 + wr * 128 + l15)) * k + g4 * 8;
    auto load_half = [&](s16x8_t(&f)[4][2], int h, int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                f[i][ks] = *(const s16x8_t *)(row0 + (size_t)((h * 4 + i) * 16) * k + kt * 64 + ks * 32);
    };
    auto mul_half = [&](const s16x8_t(&f)[4][2], int h) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[h * 4 + i][j] = Tr::mfma(b[j][ks], f[i][ks], acc[h * 4 + i][j]);
    };
    const int nkt = k / 64;
    s16x8_t f0[4][2], f1[4][2];
    load_half(f0, 0, 0);
    int kt = 0;
    for (int it = 0; it < iters; ++it) {
        load_half(f1, 1, kt);
        mul_half(f0, 0);
        const int nx = kt + 1 == nkt ? 0 : kt + 1;
        load_half(f0, 0, nx);
        mul_half(f1, 1);
        kt = nx;
    }
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            s += acc[i][j];
    if (s[0] + s[1] + s[2] + s[3] == 123.456f)
        sink[blockIdx.x * 512 + t] = s[0];
}

// a: [2048][k] 16-bit operands (k % 64 == 0, k >= 64; 2048 x 512 = 2 MB stays in every XCD's L2), bdata: >= 16 * 512 * 8 * 16 bytes,
// sink: >= num_cu * 512 floats. One launch of num_cu workgroups x 512 threads, iters K-tiles (64 MFMAs + 16 fragment loads per wave each).
extern "C" int infini_rocm_probe_mfma_a_from_l2(infiniRocmRuntime_t rt, int dtype, const void *a, const void *bdata, void *sink, int k, int iters,
                                                double *flop) {
    IROCM_CHECK_ARG(rt && a && bdata && sink && iters > 0 && k >= 64 && k % 64 == 0, "probe: bad argument");
    IROCM_CHECK_ARG(dtype == INFINI_DT_BF16 || dtype == INFINI_DT_F16, "probe: bf16 / f16 only");
    const unsigned grid = (unsigned)rt->num_cu;
    if (dtype == INFINI_DT_BF16)
        hipLaunchKernelGGL(mfma_a_from_l2_kernel<Bf16Traits>, dim3(grid), dim3(512), 0, rt->stream, (const unsigned short *)a,
                           (const unsigned short *)bdata, (float *)sink, k, iters);
    else
        hipLaunchKernelGGL(mfma_a_from_l2_kernel<F16Traits>, dim3(grid), dim3(512), 0, rt->stream, (const unsigned short *)a,
                           (const unsigned short *)bdata, (float *)sink, k, iters);
    IROCM_LAUNCH_CHECK("mfma_a_from_l2");
    if (flop)
        *flop = (double)grid * 8.0 * (double)iters * 64.0 * (2.0 * 16 * 16 * 32);
    return INFINI_ROCM_OK;
}

// data: >= 16 * 512 * 12 * 16 bytes (1.5 MiB) of 16-bit operands in device memory (random data = the realistic power
// draw; zeros clock higher); sink: >= num_cu * 512 floats. Launches ONE kernel of num_cu workgroups x 512 threads that
// issues iters * 64 MFMAs per wave; *flop receives the FLOP count of the launch (time it with events).
extern "C" int infini_rocm_probe_mfma_ceiling(infiniRocmRuntime_t rt, int dtype, const void *data, void *sink, int iters,
                                              double *flop) {
    IROCM_CHECK_ARG(rt && data && sink && iters > 0, "probe: bad argument");
    IROCM_CHECK_ARG(dtype == INFINI_DT_BF16 || dtype == INFINI_DT_F16, "probe: bf16 / f16 only");
    const unsigned grid = (unsigned)rt->num_cu;
    if (dtype == INFINI_DT_BF16)
        hipLaunchKernelGGL(mfma_ceiling_kernel<Bf16Traits>, dim3(grid), dim3(512), 0, rt->stream, (const unsigned short *)data,
                           (float *)sink, iters);
    else
        hipLaunchKernelGGL(mfma_ceiling_kernel<F16Traits>, dim3(grid), dim3(512), 0, rt->stream, (const unsigned short *)data,
                           (float *)sink, iters);
    IROCM_LAUNCH_CHECK("mfma_ceiling");
    if (flop)
        *flop = (double)grid * 8.0 * (double)iters * 64.0 * (2.0 * 16 * 16 * 32);
    return INFINI_ROCM_OK;
}
