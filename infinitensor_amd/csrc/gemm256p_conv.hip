// gemm256p in conv mode (gemm256p_kernel.h, CONV): pointwise convolutions with >= 256 filters as ONE GEMM over pixel slots.
// One translation unit for the three tile widths (A K-major, B gathered from NCHW: one layout each).
#include "gemm256p_kernel.h"

namespace irocm {
namespace g256p {

// p: m = F, n = images * hwp, k = C, a = W [F][C], b = X (NCHW), c = Y (NCHW), bias = [F] or nullptr, cv_* set by the caller.
int launch_gemm256p_conv(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, int nt) {
    const bool res = p.cv_res != nullptr;
    if (dtype == INFINI_DT_BF16) {
        if (res)
            return nt == 4 ? launch_p_conv<Bf16Traits, 4, true>(rt, p)
                           : (nt == 3 ? launch_p_conv<Bf16Traits, 3, true>(rt, p) : launch_p_conv<Bf16Traits, 2, true>(rt, p));
        if (nt == 4) return launch_p_conv<Bf16Traits, 4, false>(rt, p);
        if (nt == 3) return launch_p_conv<Bf16Traits, 3, false>(rt, p);
        return launch_p_conv<Bf16Traits, 2, false>(rt, p);
    }
    if (res)
        return nt == 4 ? launch_p_conv<F16Traits, 4, true>(rt, p)
                       : (nt == 3 ? launch_p_conv<F16Traits, 3, true>(rt, p) : launch_p_conv<F16Traits, 2, true>(rt, p));
    if (nt == 4) return launch_p_conv<F16Traits, 4, false>(rt, p);
    if (nt == 3) return launch_p_conv<F16Traits, 3, false>(rt, p);
    return launch_p_conv<F16Traits, 2, false>(rt, p);
}

} // namespace g256p

int persist_pick_nt(long m, long n, long k, int cus, int max_nt);

// Pointwise convolution as one GEMM over pixel slots (see infini_rocm_conv2d_res). Returns -1 when the operands do not qualify
// (the caller falls through to the other kernels), a status otherwise.
int launch_conv_pw_gemm(infiniRocmRuntime_t rt, int dtype, const void *x, const void *w, const void *bias, const void *res,
                               void *y, int64_t n, int64_t c, int64_t hw, int64_t f, int act) {
    const int64_t hwp = (hw + 7) & ~(int64_t)7;
    if (hw < 8 || (((uintptr_t)w) & 15) != 0 || (((uintptr_t)x) & 1) != 0 || (((uintptr_t)y) & 1) != 0)
        return -1;
    // 32-bit byte offsets into X for the LDS-DMA, 32-bit element offsets into Y / the residual
    if (n * c * hw * 2 >= (1ll << 32) - 64 || n * f * hw >= (1ll << 31) - 64 || n * hwp >= (1ll << 31) - 512)
        return -1;
    if (hw % 8 != 0) {
        // the last 16-byte run of a plane reaches up to 14 bytes past it: at the very end of X that must be readable memory.
        // True when the allocation X lives in goes on (an arena of infini_rocm_alloc always does: 256 bytes of slack; a caller's own
        // tensor that ends exactly at the end of its hipMalloc block does not, and takes the other kernels).
        const char *x_end = (const char *)x + n * c * hw * 2;
        hipDeviceptr_t base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)(x_end - 2)) != hipSuccess) {
            (void)hipGetLastError();
            return -1;
        }
        if (x_end + 16 > (const char *)base + size)
            return -1;
    }
    GemmArgs p;
    memset(&p, 0, sizeof(p));
    p.a = w; p.b = x; p.bias = bias; p.c = y;
    p.m = (int)f; p.n = (int)(n * hwp); p.k = (int)c; p.batch = 1;
    p.a_rs = c; p.a_cs = 1;
    p.act = act;
    p.splitk = 1;
    p.zeros = rt->zeros;
    p.epi16 = 1;
    p.cv_hw = (int)hw; p.cv_hwp = (int)hwp; p.cv_res = res;
    p.cv_res_bytes = (unsigned)(n * f * hw * 2); // (< 2^32: checked above)
    // the 256-column residual copy spills 36-48 bytes per lane (epilogue only) and still wins where the cost model picks it:
    // C256 -> F1024 @14x14 with a residual 40.4 vs 46.9 us on 192-column tiles, C512 -> F2048 @7x7 30.7 vs 36.6 (IROCM_CONV_RES_NT4=0: A/B)
    static const int res_nt4 = getenv("IROCM_CONV_RES_NT4") ? atoi(getenv("IROCM_CONV_RES_NT4")) : 1;
    int nt = persist_pick_nt(f, n * hwp, c, rt->num_cu, (res && !res_nt4) ? 3 : 4);
    // The cost model is fitted on MFMA-bound GEMMs, where a partial last round of tiles costs most of a full one. Layers with C <= 512
    // are HBM-bound (4-8 K-tiles per tile): a partial round simply has the bandwidth to itself, and narrower tiles only add tile
    // boundaries — measured at batch 128 (tools/gpu_r5k.sh, us with the model's width / 256 columns): C512 -> 128 @28^2 39.4 / 34.8,
    // C512 -> 256 @28^2 40.6 / 37.5, C128 -> 512 @28^2 40.9 / 34.3; with C >= 1024 the model's narrower tiles stay (C1024 -> 256 @14^2
    // 23.0 / 27.5). The residual form at C = 64 (ONE K-tile per tile: all epilogue) is the short-K case that likes 128 columns —
    // half the residual registers in flight per wave, twice the tiles to balance (103.3 / 109.2).
    if (c <= 512)
        nt = (res && c <= 64) ? 2 : 4;
    if (const char *force = getenv("IROCM_CONV_PW_NT")) { // test hook (read per call): force the tile width 2 / 3 / 4
        const int v = atoi(force);
        if (v >= 2 && v <= 4)
            nt = v;
    }
    rt->last_conv_route = "pixel_gemm";
    return g256p::launch_gemm256p_conv(rt, dtype, p, nt);
}

} // namespace irocm
