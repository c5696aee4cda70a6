// gemm256p in TAP mode (gemm256p_kernel.h, CONV = 3): 3 x 3 convolutions (pad 1, dilation 1, stride 1 or 2) with >= 256 filters as
// ONE GEMM over pixel slots with K = 9 C on the persistent 256-row kernels. Reference semantics: src/kernels/cuda/conv.cc:57-168
// (cuDNN cross-correlation, NCHW x FCRS), output extent src/operators/conv.cc:98-101.
#include "gemm256p_kernel.h"

namespace irocm {
namespace g256p {

int launch_gemm256p_conv3(infiniRocmRuntime_t rt, int dtype, const GemmArgs &p, int nt, int split, char *slab, size_t slab_bytes) {
    // timeline build (tools/conv_tap_timeline.py): IROCM_CONV_TAP_TRACE = device address (hex) of [grid][8][128] uint64 stamps
    if (const char *tr = getenv("IROCM_CONV_TAP_TRACE")) {
        unsigned long long *trace = (unsigned long long *)strtoull(tr, nullptr, 16);
        if (trace && dtype == INFINI_DT_F16 && nt == 4)
            return launch_p_conv<F16Traits, 4, false, true, true>(rt, p, split, slab, slab_bytes, rt->sync_flags, trace);
    }
    if (split > 1) // (the split form exists for 256-column tiles: it is chosen when the 256 x 256 tiles alone cannot fill the chip)
        return dtype == INFINI_DT_BF16 ? launch_p_conv<Bf16Traits, 4, false, true>(rt, p, split, slab, slab_bytes, rt->sync_flags)
                                       : launch_p_conv<F16Traits, 4, false, true>(rt, p, split, slab, slab_bytes, rt->sync_flags);
    if (dtype == INFINI_DT_BF16) {
        if (nt == 4) return launch_p_conv<Bf16Traits, 4, false, true>(rt, p);
        if (nt == 3) return launch_p_conv<Bf16Traits, 3, false, true>(rt, p);
        return launch_p_conv<Bf16Traits, 2, false, true>(rt, p);
    }
    if (nt == 4) return launch_p_conv<F16Traits, 4, false, true>(rt, p);
    if (nt == 3) return launch_p_conv<F16Traits, 3, false, true>(rt, p);
    return launch_p_conv<F16Traits, 2, false, true>(rt, p);
}

} // namespace g256p

int persist_pick_nt(long m, long n, long k, int cus, int max_nt);

// Split-K of the tap GEMM: how many workgroups share one 256 x 256 tile (1 = no split) and the bytes of the fp32 exchange slab the
// caller has to provide. The split form runs ONE unit per workgroup (grid <= CUs: every slice of every tile resident at once), so
// it is taken only when the 256 x 256 tiles number less than half the CUs — ResNet-50 at batch 128: 100 tiles (14 x 14 planes, F 256)
// -> 2 slices of 18 K-tiles, 56 tiles (7 x 7, F 512) -> 4 slices of 18.
int conv_tap_split(infiniRocmRuntime_t rt, int64_t n, int64_t hw, int64_t c, int64_t f, size_t *slab_bytes) {
    const int64_t hwp = (hw + 7) & ~(int64_t)7;
    const int64_t tiles = ceil_div(f, 256) * ceil_div(n * hwp, 256);
    const int64_t nk = 9 * (c / 64);
    const int cus = rt->num_cu >= 8 ? (rt->num_cu / 8) * 8 : rt->num_cu;
    int split = 1;
    const int forced = getenv("IROCM_CONV_TAP_SPLIT") ? atoi(getenv("IROCM_CONV_TAP_SPLIT")) : 0; // test / A-B hook (read per call): 1 = never split, 2 / 4 = only that factor
    for (int s : {4, 2}) {
        // every slice keeps >= 9 K-tiles (one channel block's taps); the flag words of all (tile, source, destination, wave) fit
        if (nk % s == 0 && nk / s >= 9 && ceil_div(tiles, 8) * 8 * s <= cus && tiles * s * s * 8 <= (int64_t)infiniRocmRuntime::kSyncFlagWords &&
            (forced == 0 || forced == s)) {
            split = s;
            break;
        }
    }
    if (forced == 1)
        split = 1;
    if (slab_bytes)
        *slab_bytes = split > 1 ? (size_t)tiles * split * 8 * 8 * 4 * 1024 : 0;
    return split;
}

// x: the activation the taps address — the layer's input X [n][c][oh][ow] for a unit-stride layer, or the four phase planes
// [py * 2 + px][n][c][oh][ow] of a stride-2 layer (conv_s1.hip's phase split, slot order py * 2 + px). wp: the weights re-packed
// [tap][F][C]. oh x ow: the OUTPUT plane (= a phase plane; = the input plane at unit stride). in_h x in_w: the layer's input extent
// (validity of a tap: 0 <= oy * stride - 1 + r < in_h). `front_ok`: the caller has proved that the bytes the moved runs reach in front
// of and behind the activation are readable. Returns -1 when the operands do not qualify (the caller takes the other kernels).
int launch_conv_tap_gemm(infiniRocmRuntime_t rt, int dtype, const void *x, const void *wp, const void *bias, void *y, int64_t n,
                         int64_t c, int oh, int ow, int in_h, int in_w, int stride, int64_t plane_elems, int64_t f, int act,
                         int split, void *slab, size_t slab_bytes) {
    const int64_t hw = (int64_t)oh * ow, hwp = (hw + 7) & ~(int64_t)7;
    if (hw < 8 || c % 64 != 0 || (((uintptr_t)wp) & 15) != 0 || (((uintptr_t)x) & 1) != 0 || (((uintptr_t)y) & 1) != 0)
        return -1;
    if (!(stride == 1 || stride == 2) || !(act == 0 || act == 1))
        return -1;
    // 32-bit byte offsets into the activation (all phase planes) for the LDS-DMA, 32-bit element offsets into Y
    const int64_t planes = stride == 1 ? 1 : 4;
    if (planes * plane_elems * 2 >= (1ll << 31) - 4096 || n * f * hw >= (1ll << 31) - 64 || n * hwp >= (1ll << 31) - 512 ||
        f * c * 2 * 9 >= (1ll << 31))
        return -1;
    GemmArgs p;
    memset(&p, 0, sizeof(p));
    p.a = wp; p.b = x; p.bias = bias; p.c = y;
    p.m = (int)f; p.n = (int)(n * hwp); p.k = (int)(9 * c); p.batch = 1;
    p.a_rs = c; p.a_cs = 1;
    p.act = act;
    p.splitk = 1;
    p.zeros = rt->zeros;
    p.epi16 = 1;
    p.cv_hw = (int)hw; p.cv_hwp = (int)hwp; p.cv_res = nullptr; p.cv_res_bytes = 0;
    p.cv_taps = 9;
    p.cv_ow = ow;
    p.cv_ow_m = udiv_magic((unsigned long long)ow);
    p.cv_atap = (int)(f * c * 2);
    // tap (r, s) reads input pixel (oy * stride - 1 + r, ox * stride - 1 + s): inside the image for
    //   r = 0: oy * stride >= 1;  r = 2: oy * stride + 1 <= in_h - 1  (r = 1 always: oy * stride <= in_h - 1 by the output extent)
    p.cv_ylo = 1;                                  // (stride 1 and 2 alike: oy >= 1)
    p.cv_yhi = (in_h - 2) / stride + 1;            // oy <= (in_h - 2) / stride
    p.cv_xlo = 1;
    p.cv_xhi = (in_w - 2) / stride + 1;
    // byte offset of tap (r, s) relative to the pointwise tile: rowoff[r] + coloff[s]
    int64_t rowoff[3], coloff[3];
    if (stride == 1) {
        for (int t = 0; t < 3; ++t) {
            rowoff[t] = (int64_t)(t - 1) * ow * 2;
            coloff[t] = (int64_t)(t - 1) * 2;
        }
    } else {
        // input row 2 oy - 1 + r = 2 (oy + dy) + py: r = 0 -> (py 1, dy -1), r = 1 -> (py 0, dy 0), r = 2 -> (py 1, dy 0); plane slot = py * 2 + px
        const int py[3] = {1, 0, 1}, dy[3] = {-1, 0, 0};
        for (int t = 0; t < 3; ++t) {
            rowoff[t] = ((int64_t)py[t] * 2 * plane_elems + (int64_t)dy[t] * ow) * 2;
            coloff[t] = ((int64_t)py[t] * plane_elems + (int64_t)dy[t]) * 2;
        }
    }
    p.cv_b0 = (int)(rowoff[0] + coloff[0]);
    p.cv_ds01 = (int)(coloff[1] - coloff[0]);
    p.cv_ds12 = (int)(coloff[2] - coloff[1]);
    p.cv_dr01 = (int)(rowoff[1] - rowoff[0] + coloff[0] - coloff[2]);
    p.cv_dr12 = (int)(rowoff[2] - rowoff[1] + coloff[0] - coloff[2]);
    p.cv_dcb = (int)(64 * hw * 2 + rowoff[0] - rowoff[2] + coloff[0] - coloff[2]);
    if (split > 1 && (!slab || slab_bytes >= (1ull << 31) || !rt->sync_flags))
        split = 1;
    int nt = split > 1 ? 4 : persist_pick_nt(f, n * hwp, 9 * c, rt->num_cu, 4);
    if (const char *force = getenv("IROCM_CONV_TAP_NT")) { // test / tuning hook (read per call): force the tile width 2 / 3 / 4
        const int v = atoi(force);
        if (v >= 2 && v <= 4) {
            nt = v;
            split = 1; // (a forced width measures the unsplit form)
        }
    }
    rt->last_conv_route = split > 1 ? "tap_gemm_splitk" : "tap_gemm";
    return g256p::launch_gemm256p_conv3(rt, dtype, p, nt, split, (char *)slab, slab_bytes);
}

} // namespace irocm
