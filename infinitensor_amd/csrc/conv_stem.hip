// The stem of a CNN as ONE kernel: Conv2d(7 x 7, stride 2, pad 3, C = 3 -> F = 64) + per-filter bias + ReLU + MaxPool(3 x 3,
// stride 2, pad 1), f16 / bf16, NCHW — ResNet-50's first four operators after the front-end's lowering
// (Conv -> Reshape(bias) -> Add -> Relu -> MaxPool; reference kernels: conv.cc:57-168 (cuDNN), element_wise.cu, unary.cc,
// pooling.cc:6-95). Round 3 ran them as a tap-shifted implicit GEMM with a per-k offset table (171 us at batch 128, 4.4 x its
// floor, behind a phase-split pre-pass) plus a pooling pass (52 us) that re-read the 205 MB the conv had just written.
//
// Here the conv tile never leaves the CU: a workgroup owns 2 x 28 POOLED outputs of one image (all 64 filters), computes the
// 5 x 57 conv pixels they need on the matrix cores, parks them (bias added, rounded, ReLU'd — the values the unfused chain
// would have stored) in LDS and pools from there; HBM sees the input once (+ halo rows from L2) and the pooled output once.
//   * GEMM view: M = 64 filters, N = conv pixels, K = (c, r, s8) with the 7 taps of a filter row padded to 8, so that a lane's B
//     fragment of v_mfma_f32_16x16x32 — 8 consecutive k of one pixel — is 8 CONSECUTIVE input pixels of one (channel, input
//     row): one 16-byte read of the input tile in LDS at a 4-byte-aligned address (the tile's origin is the odd input column
//     2 x0 - 3, so a run starts on an even element). 21 (c, r) pairs -> 6 k-steps of 4 pairs (3 zero-weight pairs).
//   * A = the weights, re-packed ONCE (packed-weight cache) into fragment order [k-step][filter tile][lane][8]: a lane keeps all
//     24 fragments (96 VGPRs) for the whole workgroup.
//   * persistent workgroups (two per CU) walk the tiles: weights and bias are fetched once per workgroup, not once per tile
//     (the first version re-read its 96 KB of fragments per tile and took 163 us).
//   * input tile: 3 channels x 15 rows x 120 columns, pitch 160 elements (row stride = 16 banks: the four (c, r) pairs a wave
//     reads at once hit disjoint banks); conv tile for the pool: [row][column][filter] with a pixel pitch of 72 elements — a
//     lane's four consecutive filters of a pixel are ONE conflict-free 8-byte LDS store, and a pooled output's 3 x 3 window
//     of 8 filters is nine aligned 16-byte reads reduced by v_pk_max_u16 (ReLU made everything >= 0: unsigned 16-bit order =
//     float order for f16 and bf16, and padding with 0 equals MaxPool's -inf padding).
// Served: C = 3, F = 64, 7 x 7 / 2 / pad 3 / dilation 1, groups 1, activation ReLU, pool 3 x 3 / 2 / pad 1 / dilation 1 /
// floor mode, W % 8 == 0, 16-byte aligned tensors. Everything else keeps the separate kernels (the query below says which).
// A zero weight multiplies the 8th element of every run: an Inf / NaN pixel therefore poisons the conv pixel to its LEFT as
// well (its true window ends one column earlier) — finite images, i.e. every real one, are unaffected.
#include "gemm_common.h"

namespace irocm {

constexpr int kStemF = 64;          // filters
constexpr int kStemQ = 24;          // (c, r) pairs incl. 3 padding pairs
constexpr int kStemKS = 6;          // k-steps of 32
constexpr int kStemPR = 2, kStemPC = 28; // pooled rows / columns per workgroup
constexpr int kStemCR = 2 * kStemPR + 1, kStemCC = 2 * kStemPC + 1; // conv rows / columns per workgroup: 5 x 57
constexpr int kStemIR = 2 * kStemCR + 5; // input rows: 15
constexpr int kStemPitch = 160;     // input tile row pitch (elements)
constexpr int kStemMargin = 8;      // elements in front of a tile row (see the staging loop); 8 + 134 < 160
constexpr int kStemInElems = 3 * kStemIR * kStemPitch; // 7200
constexpr int kStemPix = 72;        // conv tile: elements per pixel (64 filters + 8 pad: 36 dwords -> conflict-free b64 stores)
constexpr int kStemConvElems = kStemCR * kStemCC * kStemPix; // 20,520
constexpr int kStemLds = (kStemInElems + kStemConvElems) * 2; // 55,440 B: two workgroups per CU

struct StemArgs {
    const unsigned short *x;   // [n][3][h][w]
    const unsigned short *wp;  // packed weights: [6 k-steps][4 filter tiles][64 lanes][8]
    const unsigned short *bias; // [64] or nullptr
    unsigned short *y;         // [n][64][ph][pw]
    int n, h, w, oh, ow, ph, pw;
    int tiles_r, tiles_c;      // pooled row / column tiles per image
    int xcd_runs;              // 1: each XCD takes a contiguous run of tiles per step (see the tile loop)
};

// FCRS [64][3][7][7] -> fragment order. Element e of lane (g4, l15) of fragment (ks, mt): filter mt * 16 + l15,
// pair q = 4 ks + g4 -> (c, r) = (q / 7, q % 7) for q < 21, tap s = e (< 7); zero otherwise.
__global__ __launch_bounds__(256) void stem_pack_w_kernel(const unsigned short *__restrict__ w, unsigned short *__restrict__ wp) {
    const int i = blockIdx.x * 256 + threadIdx.x; // one element
    if (i >= kStemKS * 4 * 64 * 8)
        return;
    const int e = i & 7, lane = (i >> 3) & 63, mt = (i >> 9) & 3, ks = i >> 11;
    const int f = mt * 16 + (lane & 15), q = 4 * ks + (lane >> 4);
    unsigned short v = 0;
    if (q < 21 && e < 7)
        v = w[((f * 3 + q / 7) * 7 + q % 7) * 7 + e];
    wp[i] = v;
}

template <typename Tr>
__global__ __launch_bounds__(256, 2) void conv_stem_pool_kernel(StemArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    unsigned short *tin = lds;                  // [3][15][160]
    unsigned short *tcv = lds + kStemInElems;   // [5][57][72]: conv + bias, rounded, ReLU'd
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, l15 = lane & 15, g4 = lane >> 4;

    // ---- weights: 24 fragments per lane, resident for every tile this workgroup walks ----------------------------------
    s16x8_t af[kStemKS][4];
#pragma unroll
    for (int ks = 0; ks < kStemKS; ++ks)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            af[ks][mt] = *(const s16x8_t *)(p.wp + ((ks * 4 + mt) * 64 + lane) * 8);
    float bv[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            bv[mt][r] = p.bias ? Tr::to_f32(p.bias[mt * 16 + 4 * g4 + r]) : 0.f;
    int qoff[kStemKS]; // element offset of pair q = 4 ks + g4 inside the tile (padding pairs read pair 0: finite data x weight 0)
#pragma unroll
    for (int ks = 0; ks < kStemKS; ++ks) {
        const int q = 4 * ks + g4;
        qoff[ks] = q < 21 ? ((q / 7) * kStemIR + q % 7) * kStemPitch : 0;
    }
    const int xl = wv * 16 + l15; // conv column inside the tile (0 .. 63; 57 used)
    const int ntiles = p.n * p.tiles_r * p.tiles_c;

    // The input tile of the NEXT tile is fetched into registers (3 x 16 bytes per thread) while this tile computes: the first
    // version loaded it at the top of every tile and idled out the HBM round trip (6.5 us per tile).
    // 16-byte chunks of the image rows: chunk m holds input columns 8 m .. 8 m + 7; a tile needs columns ix0 .. ix0 + 119,
    // i.e. the 16 chunks from floor(ix0 / 8) on; chunks / rows outside the image are zeros (the convolution's padding).
    constexpr int kItems = 3 * kStemIR * 16, kPer = (kItems + 255) / 256; // 720 items, 3 per thread
    s16x8_t pre[kPer];
    auto decode = [&](int tile, int &img, int &p0, int &c0) {
        const int tc = tile % p.tiles_c;
        const int rest = tile / p.tiles_c;
        p0 = (rest % p.tiles_r) * kStemPR;
        c0 = tc * kStemPC;
        img = rest / p.tiles_r;
    };
    auto fetch = [&](int tile) {
        int img, p0, c0;
        decode(tile, img, p0, c0);
        const int iy0 = 2 * (2 * p0 - 1) - 3, ix0 = 2 * (2 * c0 - 1) - 3;
        const int m_first = ix0 >= 0 ? ix0 / 8 : -1; // floor(ix0 / 8): ix0 = -5 for the first column tile, positive afterwards
        const unsigned short *X = p.x + (long)img * 3 * p.h * p.w;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int i = t + j * 256;
            const int m = m_first + (i & 15), rowc = i >> 4; // rowc = c * 15 + row
            const int c = rowc / kStemIR, iy = iy0 + rowc - c * kStemIR, ix = 8 * m;
            const bool ok = i < kItems && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
            pre[j] = s16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
            if (ok)
                pre[j] = *(const s16x8_t *)(X + ((long)c * p.h + iy) * p.w + ix);
        }
    };
    // Tile order (round 5): workgroup b runs on XCD b % 8. With tile = b + step * grid, vertically adjacent tiles of an image (t and
    // t + tiles_c) sat on DIFFERENT XCDs: the 7 input rows two row tiles share were fetched from HBM once per XCD (FETCH 99 MB for a
    // 38.5 MB input) and the 128-byte lines of the pooled output that straddle two tiles (rows of 112 bytes) were written partially by
    // two L2s (WRITE 71.7 MB for 51.4 MB). Each XCD now takes a CONTIGUOUS run of grid / 8 tiles per step (a little more than one
    // image): halos and shared lines meet in one L2.
    const int first = (p.xcd_runs && (gridDim.x & 7) == 0) ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (first < ntiles)
        fetch(first);

    for (int tile = first; tile < ntiles; tile += gridDim.x) {
        int img, p0, c0;
        decode(tile, img, p0, c0);
        const int y0 = 2 * p0 - 1, x0 = 2 * c0 - 1;      // first conv row / column of the tile
        const int ix0 = 2 * x0 - 3;                      // first input column of the tile (odd)

        // ---- input tile: registers -> LDS (every element a live pixel reads is written: no zero fill) -----------------------
        {
            const int m_first = ix0 >= 0 ? ix0 / 8 : -1;
#pragma unroll
            for (int j = 0; j < kPer; ++j) {
                const int i = t + j * 256;
                if (i < kItems) {
                    // odd element offset (-7 .. 120): element stores; a row starts kStemMargin elements into its pitch, so the
                    // first chunk's elements left of the tile land in the margin (never read) instead of needing a guard
                    const int rowc = i >> 4, off = 8 * (m_first + (i & 15)) - ix0;
                    unsigned short *dst = tin + rowc * kStemPitch + kStemMargin + off;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        dst[e] = (unsigned short)pre[j][e];
                }
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles)
            fetch(tile + gridDim.x); // in flight during the conv and pool phases below

        // ---- conv tile: wave wv takes the 16-column block wv of each of the 5 conv rows ---------------------------------
#pragma unroll 1
        for (int yi = 0; yi < kStemCR; ++yi) {
            const unsigned short *base = tin + 2 * yi * kStemPitch + kStemMargin + 2 * xl;
            u32x4_t bq[kStemKS];
#pragma unroll
            for (int ks = 0; ks < kStemKS; ++ks) { // all six fragments first: the LDS latency is paid once per pixel tile
                const unsigned *src = (const unsigned *)(base + qoff[ks]); // 4-byte aligned
                bq[ks] = u32x4_t{src[0], src[1], src[2], src[3]};
            }
            f32x4 acc[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < kStemKS; ++ks) {
                const s16x8_t bf = __builtin_bit_cast(s16x8_t, bq[ks]);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[mt] = Tr::mfma(af[ks][mt], bf, acc[mt]);
            }
            // bias, ReLU, round -> [row][column][filter]: the lane's filters mt * 16 + 4 g4 .. + 3 of its pixel are 8 contiguous bytes.
            // round(relu(v)) == relu(round(v)): the bits the unfused Conv + Add -> Relu chain stores. Pixels outside the conv
            // output are 0 (= MaxPool's padding after a ReLU); columns 57 .. 63 of the block are never read.
            const int gy = y0 + yi, gx = x0 + xl;
            const bool live = gy >= 0 && gy < p.oh && gx >= 0 && gx < p.ow;
            if (xl < kStemCC) {
                unsigned short *dst = tcv + (yi * kStemCC + xl) * kStemPix + 4 * g4;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc[mt][r] + bv[mt][r];
                        v[r] = (live && v[r] > 0.f) ? v[r] : 0.f;
                    }
                    *(u32x2_t *)(dst + mt * 16) = u32x2_t{Tr::pack2(v[0], v[1]), Tr::pack2(v[2], v[3])};
                }
            }
        }
        __syncthreads();

        // ---- pool: (pooled row, pooled column, 8 filters): nine 16-byte reads, packed unsigned max, 8 element stores (lanes run
        // along the pooled columns: 56 contiguous bytes per store instruction). A form with 8-byte stores — a thread pooling four
        // adjacent columns of four filters from 27 8-byte reads — was measured SLOWER (110 vs 86 us at batch 128): the pool phase
        // is bound by its per-thread instruction count, not by the 2-byte stores.
        typedef unsigned short u16x8_t __attribute__((ext_vector_type(8)));
        for (int i = t; i < kStemPR * kStemPC * 8; i += 256) {
            const int pcl = i % kStemPC, rest = i / kStemPC, pr = rest % kStemPR, f8 = rest / kStemPR;
            const int prow = p0 + pr, pc = c0 + pcl;
            if (prow >= p.ph || pc >= p.pw)
                continue;
            u16x8_t m = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
                    m = __builtin_elementwise_max(m, *(const u16x8_t *)(tcv + ((2 * pr + dy) * kStemCC + 2 * pcl + dx) * kStemPix + f8 * 8));
            unsigned short *dst = p.y + (((long)img * kStemF + f8 * 8) * p.ph + prow) * p.pw + pc;
            const long fstride = (long)p.ph * p.pw;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                dst[e * fstride] = m[e];
        }
        // (the next tile's staging touches `tin` only; its barrier orders the next conv phase against these reads of `tcv`)
    }
}

static bool stem_pool_shape_ok(int dtype, int c, int h, int w, int f, int r, int s, int ph, int pw, int sh, int sw, int dh, int dw,
                               int groups, int act, int pk, int ps, int pp) {
    return (dtype == INFINI_DT_F16 || dtype == INFINI_DT_BF16) && c == 3 && f == kStemF && r == 7 && s == 7 && ph == 3 && pw == 3 && sh == 2 &&
           sw == 2 && dh == 1 && dw == 1 && groups == 1 && act == 1 && pk == 3 && ps == 2 && pp == 1 && w % 8 == 0 && h >= 7 && w >= 8;
}

} // namespace irocm

using namespace irocm;

extern "C" int infini_rocm_conv2d_pool_supported(int dtype, int64_t c, int64_t h, int64_t w, int64_t f, int64_t r, int64_t s, int ph,
                                                 int pw, int sh, int sw, int dh, int dw, int groups, int act, int pool_k, int pool_s,
                                                 int pool_p) {
    return h < (1 << 20) && w < (1 << 20) &&
                   stem_pool_shape_ok(dtype, (int)c, (int)h, (int)w, (int)f, (int)r, (int)s, ph, pw, sh, sw, dh, dw, groups, act, pool_k, pool_s, pool_p)
               ? 1
               : 0;
}

extern "C" int infini_rocm_conv2d_pool(infiniRocmRuntime_t rt, int dtype, const void *x, const void *w, const void *bias, void *y,
                                       int64_t n, int64_t c, int64_t h, int64_t wd, int64_t f, int64_t r, int64_t s, int ph, int pw, int sh,
                                       int sw, int dh, int dw, int groups, int act, int pool_k, int pool_s, int pool_p) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(n >= 0, "conv2d_pool: negative batch");
    if (!infini_rocm_conv2d_pool_supported(dtype, c, h, wd, f, r, s, ph, pw, sh, sw, dh, dw, groups, act, pool_k, pool_s, pool_p) ||
        ((((uintptr_t)x) | ((uintptr_t)y)) & 15) != 0)
        IROCM_FAIL(INFINI_ROCM_UNSUPPORTED, "conv2d_pool: only the 7 x 7 / 2 stem (C = 3, F = 64, bias + ReLU) followed by MaxPool 3 x 3 / 2 / 1 on "
                                            "16-byte aligned f16 / bf16 tensors with W %% 8 == 0 is fused (ask infini_rocm_conv2d_pool_supported)");
    if (n == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(x && w && y, "conv2d_pool: NULL tensor");
    StemArgs p;
    p.n = (int)n; p.h = (int)h; p.w = (int)wd;
    p.oh = (int)((h + 6 - 7) / 2 + 1);
    p.ow = (int)((wd + 6 - 7) / 2 + 1);
    p.ph = (p.oh + 2 - 3) / 2 + 1;
    p.pw = (p.ow + 2 - 3) / 2 + 1;
    p.tiles_r = (p.ph + kStemPR - 1) / kStemPR;
    p.tiles_c = (p.pw + kStemPC - 1) / kStemPC;
    IROCM_CHECK_ARG((long)n * p.tiles_r * p.tiles_c < (1l << 31), "conv2d_pool: too many tiles");
    // packed weights: cached for graph weights (ConstWeightsScope), else built per call in the workspace
    const size_t w_bytes = (size_t)kStemKS * 4 * 64 * 8 * 2;
    const bool cached = rt->conv_const_weights != 0;
    const void *packed = nullptr;
    hipStream_t pack_stream = rt->stream;
    bool need_pack = true;
    if (cached) {
        packed = wcache_lookup(rt, w, (int)f, (int)c, (int)(r * s), 2);
        if (!packed) {
            void *buf = nullptr;
            int st = wcache_insert(rt, w, (size_t)f * c * r * s * 2, (int)f, (int)c, (int)(r * s), 2, w_bytes, &buf, &pack_stream);
            if (st != INFINI_ROCM_OK)
                return st;
            packed = buf;
        } else {
            need_pack = false;
        }
    } else {
        void *ws = nullptr;
        int st = infini_rocm_workspace(rt, w_bytes, &ws);
        if (st != INFINI_ROCM_OK)
            return st;
        packed = ws;
    }
    if (need_pack) {
        hipLaunchKernelGGL(stem_pack_w_kernel, dim3((unsigned)(w_bytes / 2 / 256)), dim3(256), 0, pack_stream, (const unsigned short *)w,
                           (unsigned short *)const_cast<void *>(packed));
        if (hipError_t e = hipGetLastError(); e != hipSuccess) {
            if (cached)
                wcache_forget(rt, packed);
            IROCM_FAIL(INFINI_ROCM_HIP_ERROR, "launch of stem_pack_w failed: %s", hipGetErrorString(e));
        }
        if (cached) {
            int st = wcache_commit(rt, pack_stream);
            if (st != INFINI_ROCM_OK) {
                wcache_forget(rt, packed);
                return st;
            }
        }
    }
    p.x = (const unsigned short *)x;
    p.wp = (const unsigned short *)packed;
    p.bias = (const unsigned short *)bias;
    p.y = (unsigned short *)y;
    const long tiles = (long)n * p.tiles_r * p.tiles_c;
    const unsigned grid = (unsigned)std::min<long>(tiles, (long)rt->num_cu * 2); // persistent: two workgroups per CU
    p.xcd_runs = getenv("IROCM_STEM_LINEAR") ? 0 : 1; // (measurement hook: the round-4 tile order)
    if (dtype == INFINI_DT_F16) {
        IROCM_LDS_ATTR(conv_stem_pool_kernel<F16Traits>, kStemLds, rt);
        hipLaunchKernelGGL(conv_stem_pool_kernel<F16Traits>, dim3(grid), dim3(256), kStemLds, rt->stream, p);
    } else {
        IROCM_LDS_ATTR(conv_stem_pool_kernel<Bf16Traits>, kStemLds, rt);
        hipLaunchKernelGGL(conv_stem_pool_kernel<Bf16Traits>, dim3(grid), dim3(256), kStemLds, rt->stream, p);
    }
    IROCM_LAUNCH_CHECK("conv_stem_pool");
    rt->last_conv_route = "stem_pool";
    return INFINI_ROCM_OK;
}
