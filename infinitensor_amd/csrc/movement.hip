// Data-movement / indexing operators for gfx950 (bit-exact, any dtype: they move raw elements of
// 1/2/4/8 bytes): Transpose, Gather, Where, Concat/Split (strided 2-D copy), Pad/Slice, Expand.
//
// Replaces (reference): TransposeCuda src/kernels/cuda/transpose.cc:8-90 + transpose.cu:10-24 (one
// thread per element, uncoalesced gather); GatherCuda gather.cc + gather.cu:4-39, include/cuda/gather.h:33-55;
// WhereCuda where.cc + where.cu:4-63; ConcatCuda/SplitCuda split_concat.cc + .cu:29-82;
// SliceCuda/PadCuda pad_slice.cc:4-45 + .cu:25-101; ExpandCuda expand.cc + expand.cu:10-49.
//
// Design: all of these are HBM-bound (bytes in + bytes out). Dimensions are collapsed on the host;
// whenever the innermost output dimension is contiguous in the input too, threads move 16-byte
// vectors; a true 2-D transpose goes through a padded 64x64 LDS tile so that both the global read
// and the global write are coalesced (the reference's transpose reads with stride).
#include "common.h"

namespace irocm {

constexpr int MD = INFINI_ROCM_MAX_DIMS;

struct IdxArgs {
    int ndim;
    long shape[MD];   // output (iteration) shape, collapsed
    long sin[MD];     // input element stride per output dim
    long total;       // number of iteration items
};

template <int BYTES> struct Raw;
template <> struct Raw<1> { using t = uint8_t; };
template <> struct Raw<2> { using t = uint16_t; };
template <> struct Raw<4> { using t = uint32_t; };
template <> struct Raw<8> { using t = uint64_t; };
template <> struct Raw<16> { using t = uint4; };

static inline unsigned grid_for(long items, int num_cu) {
    long g = ceil_div(items, 256);
    const long cap = (long)num_cu * 16;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

// out[i] = in[sum_d idx_d * sin_d]; one item = one element of BYTES bytes (a whole 16-byte vector
// when the host found the inner dimension contiguous on both sides).
struct GeArgs {
    int ndim, axis;
    long total, axis_dim;
    long idx_shape[INFINI_ROCM_MAX_DIMS], data_stride[INFINI_ROCM_MAX_DIMS];
};

template <int BYTES>
__global__ __launch_bounds__(256) void strided_gather_kernel(const void *__restrict__ in, void *__restrict__ out,
                                                             IdxArgs p) {
    using R = typename Raw<BYTES>::t;
    const R *src = (const R *)in;
    R *dst = (R *)out;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long)gridDim.x * 256) {
        long rem = i, off = 0;
        for (int d = p.ndim - 1; d >= 0; --d) {
            const long q = rem / p.shape[d];
            off += (rem - q * p.shape[d]) * p.sin[d];
            rem = q;
        }
        dst[i] = src[off];
    }
}

// Batched 2-D transpose [B, R, C] -> [B, C, R] through LDS (64x64 tile, +1 padding).
template <int BYTES>
__global__ __launch_bounds__(256) void transpose2d_kernel(const void *__restrict__ in, void *__restrict__ out,
                                                          long rows, long cols, long tiles_r, long tiles_c) {
    using R = typename Raw<BYTES>::t;
    __shared__ R tile[64][65];
    const long b = blockIdx.x / (tiles_r * tiles_c);
    const long t = blockIdx.x % (tiles_r * tiles_c);
    const long r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
    const R *src = (const R *)in + b * rows * cols;
    R *dst = (R *)out + b * rows * cols;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const long r = r0 + ty * 16 + j, c = c0 + tx;
        if (r < rows && c < cols)
            tile[ty * 16 + j][tx] = src[r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const long c = c0 + ty * 16 + j, r = r0 + tx;
        if (r < rows && c < cols)
            dst[c * rows + r] = tile[tx][ty * 16 + j];
    }
}

// The same transpose for 2- / 4-byte elements when both extents are multiples of the 16-byte vector: 16-byte global loads AND
// stores (the kernel above moves one element per lane: 2.9 TB/s on K^T of a BERT head, [512, 64] f16). A 64 x 64 tile goes
// through LDS with an 8-byte row pad; a lane gathers the V elements of its output vector from V tile rows (2-way bank
// conflicts at most) and the 8 lanes of an output row segment store 128 contiguous bytes.
template <int BYTES>
__global__ __launch_bounds__(256) void transpose2d_vec_kernel(const void *__restrict__ in, void *__restrict__ out, long rows,
                                                              long cols, long tiles_r, long tiles_c) {
    using R = typename Raw<BYTES>::t;
    constexpr int V = 16 / BYTES;      // elements per vector
    constexpr int NV = 64 / V;         // vectors per tile row
    constexpr int PITCH = 64 + 8 / BYTES; // elements
    struct alignas(16) RV { R v[V]; };
    struct alignas(8) RH { R v[V / 2]; };
    __shared__ __attribute__((aligned(16))) R tile[64 * PITCH];
    const long b = blockIdx.x / (tiles_r * tiles_c);
    const long t = blockIdx.x % (tiles_r * tiles_c);
    const long r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
    const R *src = (const R *)in + b * rows * cols;
    R *dst = (R *)out + b * rows * cols;
#pragma unroll
    for (int i = 0; i < 64 * NV / 256; ++i) {
        const int q = threadIdx.x + i * 256;
        const int r = q / NV, cv = q % NV;
        if (r0 + r < rows && c0 + cv * V < cols) {
            const RV v = *reinterpret_cast<const RV *>(src + (r0 + r) * cols + c0 + cv * V);
            RH lo, hi;
#pragma unroll
            for (int j = 0; j < V / 2; ++j) {
                lo.v[j] = v.v[j];
                hi.v[j] = v.v[V / 2 + j];
            }
            *reinterpret_cast<RH *>(&tile[r * PITCH + cv * V]) = lo;
            *reinterpret_cast<RH *>(&tile[r * PITCH + cv * V + V / 2]) = hi;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 64 * NV / 256; ++i) {
        const int q = threadIdx.x + i * 256;
        const int c = q / NV, rv = q % NV; // output row c0 + c, output columns r0 + rv * V ..
        if (c0 + c < cols && r0 + rv * V < rows) {
            RV o;
#pragma unroll
            for (int j = 0; j < V; ++j)
                o.v[j] = tile[(rv * V + j) * PITCH + c];
            *reinterpret_cast<RV *>(dst + (c0 + c) * rows + r0 + rv * V) = o;
        }
    }
}

static int launch_strided(infiniRocmRuntime_t rt, int elem, const void *in, void *out, IdxArgs p) {
    // widen the element when the innermost dim is contiguous (stride 1) and everything is aligned
    int bytes = elem;
    const int last = p.ndim - 1;
    if (p.sin[last] == 1) {
        int w = 16;
        while (w > elem) {
            const int f = w / elem;
            bool ok = (p.shape[last] % f == 0) && ((uintptr_t)in % w == 0) && ((uintptr_t)out % w == 0);
            for (int d = 0; d < last && ok; ++d)
                ok = (p.sin[d] % f == 0);
            if (ok)
                break;
            w >>= 1;
        }
        if (w > elem) {
            const int f = w / elem;
            p.shape[last] /= f;
            for (int d = 0; d < last; ++d)
                p.sin[d] /= f;
            p.total /= f;
            bytes = w;
        }
    }
    const unsigned g = grid_for(p.total, rt->num_cu);
    switch (bytes) {
    case 1: hipLaunchKernelGGL(strided_gather_kernel<1>, dim3(g), dim3(256), 0, rt->stream, in, out, p); break;
    case 2: hipLaunchKernelGGL(strided_gather_kernel<2>, dim3(g), dim3(256), 0, rt->stream, in, out, p); break;
    case 4: hipLaunchKernelGGL(strided_gather_kernel<4>, dim3(g), dim3(256), 0, rt->stream, in, out, p); break;
    case 8: hipLaunchKernelGGL(strided_gather_kernel<8>, dim3(g), dim3(256), 0, rt->stream, in, out, p); break;
    case 16: hipLaunchKernelGGL(strided_gather_kernel<16>, dim3(g), dim3(256), 0, rt->stream, in, out, p); break;
    default: IROCM_FAIL(INFINI_ROCM_UNSUPPORTED, "element size %d", elem);
    }
    IROCM_LAUNCH_CHECK("strided_gather");
    return INFINI_ROCM_OK;
}

// ---- gather_elements -----------------------------------------------------------------------------
// out[..i..] = data[.. idx[..i..] on `axis` ..]: index / output viewed as [outer, n_idx, inner], data as
// [outer, axis_dim, inner] (all other extents equal — ONNX allows index extents <= data extents off-axis; the
// general case passes per-dimension strides, this kernel covers equal off-axis extents, which is what the
// reference's tests and models use: test_cuda_gather_elements.cc).
template <int BYTES, typename I>
__global__ __launch_bounds__(256) void gather_elements_kernel(const void *__restrict__ data, const I *__restrict__ idx,
                                                              void *__restrict__ out, GeArgs p) {
    using R = typename Raw<BYTES>::t;
    const R *src = (const R *)data;
    R *dst = (R *)out;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long)gridDim.x * 256) {
        long rem = i, off = 0;
        for (int d = p.ndim - 1; d >= 0; --d) {
            const long q = rem / p.idx_shape[d];
            long c = rem - q * p.idx_shape[d];
            if (d == p.axis) {
                c = (long)idx[i];
                if (c < 0)
                    c += p.axis_dim; // ONNX negative index
            }
            off += c * p.data_stride[d];
            rem = q;
        }
        dst[i] = src[off];
    }
}

// ---- gather ------------------------------------------------------------------------------------
template <int BYTES, typename I>
__global__ __launch_bounds__(256) void gather_kernel(const void *__restrict__ data, const I *__restrict__ idx,
                                                     void *__restrict__ out, long outer, long axis_dim,
                                                     long n_idx, long inner) {
    using R = typename Raw<BYTES>::t;
    const R *src = (const R *)data;
    R *dst = (R *)out;
    const long total = outer * n_idx * inner;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long in_i = i % inner;
        const long j = (i / inner) % n_idx;
        const long o = i / (inner * n_idx);
        long k = (long)idx[j];
        if (k < 0)
            k += axis_dim; // ONNX negative index
        dst[i] = src[(o * axis_dim + k) * inner + in_i];
    }
}

// Rows of >= 16 vectors (an embedding lookup: 768 f16 = 96 x 16 B): one wave per gathered row, the index read once per row
// (wave-uniform), no per-element division (gather_kernel above does three 64-bit ones per 16 bytes).
template <typename I>
__global__ __launch_bounds__(256) void gather_rows_kernel(const void *__restrict__ data, const I *__restrict__ idx,
                                                          void *__restrict__ out, long outer, long axis_dim, long n_idx,
                                                          int inner /* 16-byte vectors per row */) {
    using R = typename Raw<16>::t;
    const R *src = (const R *)data;
    R *dst = (R *)out;
    const int lane = threadIdx.x & 63;
    const long rows = outer * n_idx;
    for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * 4) {
        const long o = outer == 1 ? 0 : r / n_idx;
        const long j = r - o * n_idx;
        long k = (long)idx[j];
        if (k < 0)
            k += axis_dim; // ONNX negative index
        const R *s = src + (o * axis_dim + k) * inner;
        R *d = dst + r * inner;
        for (int v = lane; v < inner; v += 64)
            d[v] = s[v];
    }
}

// ---- where -------------------------------------------------------------------------------------
struct WhereArgs {
    int ndim;
    long shape[MD], sx[MD], sy[MD], sc[MD];
    long total;
};

// CB = bytes per condition element; SIGNMASK clears the sign bit of a floating-point condition (-0.0 is false).
// The reference reads one byte per element (where.cu:4-19) because its CUDA Less writes bool bytes into a buffer the
// graph declares with the operands' dtype; here comparisons write full elements (like the native-CPU kernels), so the
// condition is read in ITS tensor's dtype and a Less(f32) -> Where chain works on both conventions.
template <int CB> struct CondRaw;
template <> struct CondRaw<1> { using t = uint8_t; };
template <> struct CondRaw<2> { using t = uint16_t; };
template <> struct CondRaw<4> { using t = uint32_t; };
template <> struct CondRaw<8> { using t = uint64_t; };

template <int BYTES, int CB>
__global__ __launch_bounds__(256) void where_kernel(const void *__restrict__ x, const void *__restrict__ y,
                                                    const void *__restrict__ cv, void *__restrict__ out,
                                                    WhereArgs p, unsigned long long keep) {
    using R = typename Raw<BYTES>::t;
    const typename CondRaw<CB>::t *c = (const typename CondRaw<CB>::t *)cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long)gridDim.x * 256) {
        long rem = i, ox = 0, oy = 0, oc = 0;
        for (int d = p.ndim - 1; d >= 0; --d) {
            const long q = rem / p.shape[d];
            const long id = rem - q * p.shape[d];
            ox += id * p.sx[d];
            oy += id * p.sy[d];
            oc += id * p.sc[d];
            rem = q;
        }
        ((R *)out)[i] = ((unsigned long long)c[oc] & keep) ? ((const R *)x)[ox] : ((const R *)y)[oy];
    }
}

// Flat form: after collapsing, ONE dimension in which every operand is dense (stride 1) or a scalar (stride 0) — masks and
// selects of same-shape tensors. V elements per thread, 16-byte loads / stores, no index arithmetic.
template <int BYTES, int CB>
__global__ __launch_bounds__(256) void where_flat_kernel(const void *__restrict__ xv, const void *__restrict__ yv,
                                                         const void *__restrict__ cv, void *__restrict__ out, long total, int sx,
                                                         int sy, int sc, unsigned long long keep) {
    using R = typename Raw<BYTES>::t;
    using C = typename CondRaw<CB>::t;
    constexpr int V = 16 / BYTES;
    struct alignas(16) RV { R v[V]; };
    struct alignas((V * CB) < 16 ? (V * CB) : 16) CV { C v[V]; };
    const R *x = (const R *)xv, *y = (const R *)yv;
    const C *c = (const C *)cv;
    const long nvec = total / V;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long)gridDim.x * 256) {
        RV a, b, o;
        CV m;
        if (sx) a = reinterpret_cast<const RV *>(x)[v];
        if (sy) b = reinterpret_cast<const RV *>(y)[v];
        if (sc) m = reinterpret_cast<const CV *>(c)[v];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const R av = sx ? a.v[j] : x[0], bv = sy ? b.v[j] : y[0];
            const C mv = sc ? m.v[j] : c[0];
            o.v[j] = ((unsigned long long)mv & keep) ? av : bv;
        }
        reinterpret_cast<RV *>(out)[v] = o;
    }
    if (blockIdx.x == 0)
        for (long i = nvec * V + threadIdx.x; i < total; i += 256)
            ((R *)out)[i] = ((unsigned long long)c[sc ? i : 0] & keep) ? x[sx ? i : 0] : y[sy ? i : 0];
}

// ---- pad / slice -------------------------------------------------------------------------------
struct PadSliceArgs {
    int ndim;
    long oshape[MD];  // output shape
    long ishape[MD];  // input shape
    long istride[MD]; // input element strides
    long start[MD];   // input index of output index 0 (negative = padding before)
    long step[MD];
    long total;
};

template <int BYTES>
__global__ __launch_bounds__(256) void pad_slice_kernel(const void *__restrict__ in, void *__restrict__ out,
                                                        PadSliceArgs p) {
    using R = typename Raw<BYTES>::t;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long)gridDim.x * 256) {
        long rem = i, off = 0;
        bool inb = true;
        for (int d = p.ndim - 1; d >= 0; --d) {
            const long q = rem / p.oshape[d];
            const long id = p.start[d] + (rem - q * p.oshape[d]) * p.step[d];
            inb = inb && id >= 0 && id < p.ishape[d];
            off += id * p.istride[d];
            rem = q;
        }
        R v{};
        if (inb)
            v = ((const R *)in)[off];
        ((R *)out)[i] = v;
    }
}

// ---- 2-D strided copy (Concat / Split) ---------------------------------------------------------
template <int BYTES>
__global__ __launch_bounds__(256) void copy2d_kernel(const char *__restrict__ src, char *__restrict__ dst,
                                                     long rows, long row_items, long src_pitch, long dst_pitch) {
    using R = typename Raw<BYTES>::t;
    // grid (column chunks, row groups): no per-element division
    for (long r = blockIdx.y; r < rows; r += gridDim.y) {
        const char *s = src + r * src_pitch;
        char *d = dst + r * dst_pitch;
        for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < row_items; c += (long)gridDim.x * 256)
            *(R *)(d + c * BYTES) = *(const R *)(s + c * BYTES);
    }
}

// Several 2-D strided copies of the same row count as ONE launch: blockIdx.z = segment (the inputs of a Concat / the outputs of a
// Split). One launch per segment made a two-way Split of 33 MB two 6 us kernels with a launch gap between them: 0.34 of the HBM peak
// at the config shape, 0.65 once the tensor is large enough for the gap not to matter.
constexpr int kCopySegs = 16;
struct MultiCopyArgs {
    const char *src[kCopySegs];
    char *dst[kCopySegs];
    long row_items[kCopySegs], src_pitch[kCopySegs], dst_pitch[kCopySegs];
    long rows;
};
template <int BYTES, int kCopyRowsPerTrip> __global__ __launch_bounds__(256) void copy2d_multi_kernel(MultiCopyArgs a) {
    using R = typename Raw<BYTES>::t;
    const int z = blockIdx.z;
    const char *src = a.src[z];
    char *dst = a.dst[z];
    const long items = a.row_items[z], sp = a.src_pitch[z], dp = a.dst_pitch[z];
    // Four CONSECUTIVE rows per trip, all four loads before the four stores: as a plain copy loop a thread had ONE 16-byte load in
    // flight (each store waits for its load) — 4 KB per workgroup, and the HBM-sized Concat (4 KB row segments, one column trip) sat
    // at 0.59 of the HBM peak. Rows past the end (the last group only) are loaded from the last row — no branch around a load — and
    // not stored. (Rows a grid-stride apart instead of consecutive: every thread touches four pages 16 MB apart, and short tensors
    // make every workgroup re-read the last row three times — 0.50 and 4 x slower at the config shape.) The four-row form is for
    // tensors with more row groups than the grid has rows (every workgroup then makes several trips); short tensors keep one row per
    // trip — measured at [2048, 4096]: 5.0 us with one row per trip, 18 us with four.
    for (long g = blockIdx.y; g * kCopyRowsPerTrip < a.rows; g += gridDim.y) {
        const long r0 = g * kCopyRowsPerTrip;
        for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < items; c += (long)gridDim.x * 256) {
            R v[kCopyRowsPerTrip];
#pragma unroll
            for (int u = 0; u < kCopyRowsPerTrip; ++u)
                v[u] = *(const R *)(src + (r0 + u < a.rows ? r0 + u : a.rows - 1) * sp + c * BYTES);
#pragma unroll
            for (int u = 0; u < kCopyRowsPerTrip; ++u)
                if (r0 + u < a.rows)
                    *(R *)(dst + (r0 + u) * dp + c * BYTES) = v[u];
        }
    }
}

static bool elem_ok(int e) { return e == 1 || e == 2 || e == 4 || e == 8; }

} // namespace irocm

using namespace irocm;

extern "C" {

int infini_rocm_transpose(infiniRocmRuntime_t rt, int dtype, const void *x, void *y, int ndim,
                          const int64_t *in_shape, const int *perm) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(ndim >= 0 && ndim <= MD, "transpose: rank %d out of range", ndim);
    const int elem = (int)dtype_size(dtype);
    IROCM_CHECK_ARG(elem_ok(elem), "transpose: unsupported dtype %s", dtype_name(dtype));
    long istr[MD];
    long total = 1;
    bool seen[MD] = {false};
    for (int d = ndim - 1; d >= 0; --d) {
        IROCM_CHECK_ARG(in_shape[d] >= 0, "transpose: negative extent");
        istr[d] = total;
        total *= in_shape[d];
    }
    for (int d = 0; d < ndim; ++d) {
        IROCM_CHECK_ARG(perm[d] >= 0 && perm[d] < ndim && !seen[perm[d]], "transpose: bad permutation");
        seen[perm[d]] = true;
    }
    if (total == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(x && y, "transpose: NULL tensor");
    // output dim d has extent in_shape[perm[d]] and input stride istr[perm[d]]; drop 1s, merge runs.
    IdxArgs p;
    p.ndim = 0;
    for (int d = 0; d < ndim; ++d) {
        const long e = in_shape[perm[d]], s = istr[perm[d]];
        if (e == 1)
            continue;
        if (p.ndim > 0 && p.sin[p.ndim - 1] == s * e) {
            p.shape[p.ndim - 1] *= e;
            p.sin[p.ndim - 1] = s;
        } else {
            p.shape[p.ndim] = e;
            p.sin[p.ndim] = s;
            ++p.ndim;
        }
    }
    if (p.ndim == 0) {
        p.ndim = 1; p.shape[0] = 1; p.sin[0] = 1;
    }
    p.total = total;
    if (p.ndim == 1 && p.sin[0] == 1)
        return infini_rocm_copy_inside(rt, y, x, (size_t)total * elem);
    // batched 2-D transpose: [B?, R, C] -> [B?, C, R]
    const int nd = p.ndim;
    if ((nd == 2 || nd == 3) && p.sin[nd - 1] == p.shape[nd - 2] && p.sin[nd - 2] == 1 &&
        (nd == 2 || p.sin[0] == p.shape[1] * p.shape[2])) {
        const long cols = p.shape[nd - 1], rows = p.shape[nd - 2]; // output is [.., rows_out = p.shape[nd-2], cols_out]
        // input matrix: [in_rows = cols_out][in_cols = rows_out]
        const long in_rows = cols, in_cols = rows, batch = nd == 3 ? p.shape[0] : 1;
        const long tr = ceil_div(in_rows, 64), tc = ceil_div(in_cols, 64);
        const long blocks = batch * tr * tc;
        if (blocks < (1l << 31) && (elem == 2 || elem == 4) && in_rows % (16 / elem) == 0 && in_cols % (16 / elem) == 0 &&
            ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0) {
            if (elem == 2)
                hipLaunchKernelGGL(transpose2d_vec_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, rt->stream, x, y, in_rows, in_cols, tr, tc);
            else
                hipLaunchKernelGGL(transpose2d_vec_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, rt->stream, x, y, in_rows, in_cols, tr, tc);
            IROCM_LAUNCH_CHECK("transpose2d_vec");
            return INFINI_ROCM_OK;
        }
        if (blocks < (1l << 31)) {
            switch (elem) {
            case 1: hipLaunchKernelGGL(transpose2d_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, rt->stream, x, y, in_rows, in_cols, tr, tc); break;
            case 2: hipLaunchKernelGGL(transpose2d_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, rt->stream, x, y, in_rows, in_cols, tr, tc); break;
            case 4: hipLaunchKernelGGL(transpose2d_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, rt->stream, x, y, in_rows, in_cols, tr, tc); break;
            default: hipLaunchKernelGGL(transpose2d_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, rt->stream, x, y, in_rows, in_cols, tr, tc); break;
            }
            IROCM_LAUNCH_CHECK("transpose2d");
            return INFINI_ROCM_OK;
        }
    }
    return launch_strided(rt, elem, x, y, p);
}

int infini_rocm_expand(infiniRocmRuntime_t rt, int dtype, const void *x, void *y, int ndim,
                       const int64_t *out_shape, const int64_t *x_strides) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(ndim >= 0 && ndim <= MD, "expand: rank %d out of range", ndim);
    const int elem = (int)dtype_size(dtype);
    IROCM_CHECK_ARG(elem_ok(elem), "expand: unsupported dtype %s", dtype_name(dtype));
    IdxArgs p;
    p.ndim = 0;
    long total = 1;
    for (int d = 0; d < ndim; ++d) {
        IROCM_CHECK_ARG(out_shape[d] >= 0, "expand: negative extent");
        total *= out_shape[d];
        const long e = out_shape[d], s = x_strides[d];
        if (e == 1)
            continue;
        if (p.ndim > 0 && p.sin[p.ndim - 1] == s * e) {
            p.shape[p.ndim - 1] *= e;
            p.sin[p.ndim - 1] = s;
        } else {
            p.shape[p.ndim] = e;
            p.sin[p.ndim] = s;
            ++p.ndim;
        }
    }
    if (total == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(x && y, "expand: NULL tensor");
    if (p.ndim == 0) {
        p.ndim = 1; p.shape[0] = 1; p.sin[0] = 1;
    }
    p.total = total;
    return launch_strided(rt, elem, x, y, p);
}

int infini_rocm_gather_elements(infiniRocmRuntime_t rt, int dtype, int index_dtype, const void *data,
                                const void *indices, void *y, int ndim, const int64_t *data_shape,
                                const int64_t *index_shape, int axis) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    const int elem = (int)dtype_size(dtype);
    IROCM_CHECK_ARG(elem_ok(elem), "gather_elements: unsupported dtype %s", dtype_name(dtype));
    IROCM_CHECK_ARG(index_dtype == INFINI_DT_I32 || index_dtype == INFINI_DT_I64,
                    "gather_elements: indices must be int32 or int64 (reference gather_elements.cc:33-39)");
    IROCM_CHECK_ARG(ndim >= 1 && ndim <= INFINI_ROCM_MAX_DIMS && axis >= 0 && axis < ndim, "gather_elements: bad rank / axis");
    GeArgs p;
    p.ndim = ndim;
    p.axis = axis;
    p.total = 1;
    long st = 1;
    for (int d = ndim - 1; d >= 0; --d) {
        IROCM_CHECK_ARG(index_shape[d] >= 0 && data_shape[d] >= 0, "gather_elements: negative extent");
        IROCM_CHECK_ARG(d == axis || index_shape[d] <= data_shape[d], "gather_elements: index extent exceeds data extent off-axis");
        p.idx_shape[d] = index_shape[d];
        p.data_stride[d] = st;
        st *= data_shape[d];
        p.total *= index_shape[d];
    }
    p.axis_dim = data_shape[axis];
    if (p.total == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(data && indices && y, "gather_elements: NULL tensor");
    const unsigned g = grid_for(p.total, rt->num_cu);
#define GE(B)                                                                                      \
    if (index_dtype == INFINI_DT_I64)                                                              \
        hipLaunchKernelGGL((gather_elements_kernel<B, int64_t>), dim3(g), dim3(256), 0, rt->stream, data, \
                           (const int64_t *)indices, y, p);                                        \
    else                                                                                           \
        hipLaunchKernelGGL((gather_elements_kernel<B, int32_t>), dim3(g), dim3(256), 0, rt->stream, data, \
                           (const int32_t *)indices, y, p);                                        \
    break
    switch (elem) {
    case 1: GE(1);
    case 2: GE(2);
    case 4: GE(4);
    default: GE(8);
    }
#undef GE
    IROCM_LAUNCH_CHECK("gather_elements");
    return INFINI_ROCM_OK;
}

int infini_rocm_gather(infiniRocmRuntime_t rt, int dtype, int index_dtype, const void *data,
                       const void *indices, void *y, int64_t outer, int64_t axis_dim, int64_t n_indices,
                       int64_t inner) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    const int elem = (int)dtype_size(dtype);
    IROCM_CHECK_ARG(elem_ok(elem), "gather: unsupported dtype %s", dtype_name(dtype));
    IROCM_CHECK_ARG(index_dtype == INFINI_DT_I32 || index_dtype == INFINI_DT_I64,
                    "gather: indices must be int32 or int64 (reference gather.h:33-55)");
    IROCM_CHECK_ARG(outer >= 0 && axis_dim >= 0 && n_indices >= 0 && inner >= 0, "gather: negative extent");
    const long total = outer * n_indices * inner;
    if (total == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(data && indices && y, "gather: NULL tensor");
    // widen along inner
    int bytes = elem;
    long inner_w = inner;
    for (int w = 16; w > elem; w >>= 1) {
        const int f = w / elem;
        if (inner % f == 0 && (uintptr_t)data % w == 0 && (uintptr_t)y % w == 0) {
            bytes = w;
            inner_w = inner / f;
            break;
        }
    }
    if (bytes == 16 && inner_w >= 16 && inner_w < (1l << 30)) {
        long gr = ceil_div(outer * n_indices, 4);
        if (gr > (long)rt->num_cu * 16) gr = (long)rt->num_cu * 16;
        if (index_dtype == INFINI_DT_I64)
            hipLaunchKernelGGL((gather_rows_kernel<int64_t>), dim3((unsigned)gr), dim3(256), 0, rt->stream, data, (const int64_t *)indices, y,
                               (long)outer, (long)axis_dim, (long)n_indices, (int)inner_w);
        else
            hipLaunchKernelGGL((gather_rows_kernel<int32_t>), dim3((unsigned)gr), dim3(256), 0, rt->stream, data, (const int32_t *)indices, y,
                               (long)outer, (long)axis_dim, (long)n_indices, (int)inner_w);
        IROCM_LAUNCH_CHECK("gather_rows");
        return INFINI_ROCM_OK;
    }
    const unsigned g = grid_for(outer * n_indices * inner_w, rt->num_cu);
#define GO(B)                                                                                      \
    if (index_dtype == INFINI_DT_I64)                                                              \
        hipLaunchKernelGGL((gather_kernel<B, int64_t>), dim3(g), dim3(256), 0, rt->stream, data,   \
                           (const int64_t *)indices, y, (long)outer, (long)axis_dim,               \
                           (long)n_indices, inner_w);                                              \
    else                                                                                           \
        hipLaunchKernelGGL((gather_kernel<B, int32_t>), dim3(g), dim3(256), 0, rt->stream, data,   \
                           (const int32_t *)indices, y, (long)outer, (long)axis_dim,               \
                           (long)n_indices, inner_w);                                              \
    break
    switch (bytes) {
    case 1: GO(1);
    case 2: GO(2);
    case 4: GO(4);
    case 8: GO(8);
    default: GO(16);
    }
#undef GO
    IROCM_LAUNCH_CHECK("gather");
    return INFINI_ROCM_OK;
}

int infini_rocm_where_ex(infiniRocmRuntime_t rt, int dtype, int cond_dtype, const void *x, const void *y,
                         const void *cond, void *out, int ndim, const int64_t *shape, const int64_t *stride_x,
                         const int64_t *stride_y, const int64_t *stride_c) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    const int cb = (int)dtype_size(cond_dtype);
    IROCM_CHECK_ARG(cb == 1 || cb == 2 || cb == 4 || cb == 8, "where: unsupported condition dtype %s", dtype_name(cond_dtype));
    const bool cfloat = cond_dtype == INFINI_DT_F16 || cond_dtype == INFINI_DT_BF16 || cond_dtype == INFINI_DT_F32 ||
                        cond_dtype == INFINI_DT_F64;
    const unsigned long long keep = cfloat ? ((cb == 8 ? ~0ull : ((1ull << (cb * 8)) - 1)) >> 1) : ~0ull;
    IROCM_CHECK_ARG(ndim >= 0 && ndim <= MD, "where: rank %d out of range", ndim);
    const int elem = (int)dtype_size(dtype);
    IROCM_CHECK_ARG(elem_ok(elem), "where: unsupported dtype %s", dtype_name(dtype));
    WhereArgs p;
    p.ndim = 0;
    long total = 1;
    for (int d = 0; d < ndim; ++d) {
        IROCM_CHECK_ARG(shape[d] >= 0, "where: negative extent");
        total *= shape[d];
        if (shape[d] == 1)
            continue;
        const long e = shape[d];
        if (p.ndim > 0 && p.sx[p.ndim - 1] == stride_x[d] * e && p.sy[p.ndim - 1] == stride_y[d] * e &&
            p.sc[p.ndim - 1] == stride_c[d] * e) {
            p.shape[p.ndim - 1] *= e;
            p.sx[p.ndim - 1] = stride_x[d];
            p.sy[p.ndim - 1] = stride_y[d];
            p.sc[p.ndim - 1] = stride_c[d];
        } else {
            p.shape[p.ndim] = e;
            p.sx[p.ndim] = stride_x[d];
            p.sy[p.ndim] = stride_y[d];
            p.sc[p.ndim] = stride_c[d];
            ++p.ndim;
        }
    }
    if (total == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(x && y && cond && out, "where: NULL tensor");
    if (p.ndim == 0) {
        p.ndim = 1; p.shape[0] = 1; p.sx[0] = p.sy[0] = p.sc[0] = 0;
    }
    p.total = total;
    const bool flat = p.ndim == 1 && (p.sx[0] == 0 || p.sx[0] == 1) && (p.sy[0] == 0 || p.sy[0] == 1) && (p.sc[0] == 0 || p.sc[0] == 1) &&
                      ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)cond) | ((uintptr_t)out)) & 15) == 0 && total >= 16 / elem;
    if (flat) {
        const unsigned gf = grid_for(total / (16 / elem), rt->num_cu);
#define IROCM_WHEREF(EB, CB_)                                                                                      \
    hipLaunchKernelGGL((where_flat_kernel<EB, CB_>), dim3(gf), dim3(256), 0, rt->stream, x, y, cond, out, total, (int)p.sx[0], \
                       (int)p.sy[0], (int)p.sc[0], keep)
#define IROCM_WHEREF_C(EB)                                                                                         \
    switch (cb) {                                                                                                  \
    case 1: IROCM_WHEREF(EB, 1); break;                                                                            \
    case 2: IROCM_WHEREF(EB, 2); break;                                                                            \
    case 4: IROCM_WHEREF(EB, 4); break;                                                                            \
    default: IROCM_WHEREF(EB, 8); break;                                                                           \
    }
        switch (elem) {
        case 1: IROCM_WHEREF_C(1); break;
        case 2: IROCM_WHEREF_C(2); break;
        case 4: IROCM_WHEREF_C(4); break;
        default: IROCM_WHEREF_C(8); break;
        }
#undef IROCM_WHEREF_C
#undef IROCM_WHEREF
        IROCM_LAUNCH_CHECK("where_flat");
        return INFINI_ROCM_OK;
    }
    const unsigned g = grid_for(total, rt->num_cu);
#define IROCM_WHERE(EB, CB_)                                                                      \
    hipLaunchKernelGGL((where_kernel<EB, CB_>), dim3(g), dim3(256), 0, rt->stream, x, y, cond, out, p, keep)
#define IROCM_WHERE_C(EB)                                                                         \
    switch (cb) {                                                                                 \
    case 1: IROCM_WHERE(EB, 1); break;                                                            \
    case 2: IROCM_WHERE(EB, 2); break;                                                            \
    case 4: IROCM_WHERE(EB, 4); break;                                                            \
    default: IROCM_WHERE(EB, 8); break;                                                           \
    }
    switch (elem) {
    case 1: IROCM_WHERE_C(1); break;
    case 2: IROCM_WHERE_C(2); break;
    case 4: IROCM_WHERE_C(4); break;
    default: IROCM_WHERE_C(8); break;
    }
#undef IROCM_WHERE_C
#undef IROCM_WHERE
    IROCM_LAUNCH_CHECK("where");
    return INFINI_ROCM_OK;
}

int infini_rocm_where(infiniRocmRuntime_t rt, int dtype, const void *x, const void *y, const void *cond,
                      void *out, int ndim, const int64_t *shape, const int64_t *stride_x,
                      const int64_t *stride_y, const int64_t *stride_c) {
    return infini_rocm_where_ex(rt, dtype, INFINI_DT_BOOL, x, y, cond, out, ndim, shape, stride_x, stride_y, stride_c);
}

int infini_rocm_pad_slice(infiniRocmRuntime_t rt, int dtype, const void *x, void *y, int ndim,
                          const int64_t *in_shape, const int64_t *out_shape, const int64_t *starts,
                          const int64_t *steps, int reserved) {
    (void)reserved;
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(ndim >= 0 && ndim <= MD, "pad_slice: rank %d out of range", ndim);
    const int elem = (int)dtype_size(dtype);
    IROCM_CHECK_ARG(elem_ok(elem), "pad_slice: unsupported dtype %s", dtype_name(dtype));
    PadSliceArgs p;
    p.ndim = ndim > 0 ? ndim : 1;
    long total = 1, istr = 1;
    if (ndim == 0) {
        p.oshape[0] = p.ishape[0] = 1; p.istride[0] = 1; p.start[0] = 0; p.step[0] = 1;
    }
    for (int d = ndim - 1; d >= 0; --d) {
        IROCM_CHECK_ARG(in_shape[d] >= 0 && out_shape[d] >= 0, "pad_slice: negative extent");
        p.oshape[d] = out_shape[d];
        p.ishape[d] = in_shape[d];
        p.istride[d] = istr;
        p.start[d] = starts[d];
        p.step[d] = steps ? steps[d] : 1;
        IROCM_CHECK_ARG(p.step[d] != 0, "pad_slice: zero step");
        istr *= in_shape[d];
        total *= out_shape[d];
    }
    if (total == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(y && (x || istr == 0), "pad_slice: NULL tensor");
    p.total = total;
    const unsigned g = grid_for(total, rt->num_cu);
    switch (elem) {
    case 1: hipLaunchKernelGGL(pad_slice_kernel<1>, dim3(g), dim3(256), 0, rt->stream, x, y, p); break;
    case 2: hipLaunchKernelGGL(pad_slice_kernel<2>, dim3(g), dim3(256), 0, rt->stream, x, y, p); break;
    case 4: hipLaunchKernelGGL(pad_slice_kernel<4>, dim3(g), dim3(256), 0, rt->stream, x, y, p); break;
    default: hipLaunchKernelGGL(pad_slice_kernel<8>, dim3(g), dim3(256), 0, rt->stream, x, y, p); break;
    }
    IROCM_LAUNCH_CHECK("pad_slice");
    return INFINI_ROCM_OK;
}

int infini_rocm_strided_copy(infiniRocmRuntime_t rt, const void *src, void *dst, int64_t rows,
                             int64_t row_bytes, int64_t src_pitch, int64_t dst_pitch) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(rows >= 0 && row_bytes >= 0, "strided_copy: negative extent");
    if (rows == 0 || row_bytes == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(src && dst, "strided_copy: NULL tensor");
    if (src_pitch == row_bytes && dst_pitch == row_bytes)
        return infini_rocm_copy_inside(rt, dst, src, (size_t)(rows * row_bytes));
    int w = 16;
    while (w > 1 && ((row_bytes % w) || (src_pitch % w) || (dst_pitch % w) || ((uintptr_t)src % w) ||
                     ((uintptr_t)dst % w)))
        w >>= 1;
    const long items = row_bytes / w;
    // grid: x = 256-item column chunks of a row, y = row groups (grid-stride in both): ~16 blocks per CU in all
    long gx = ceil_div(items, 256), gy = rows;
    const long cap = (long)rt->num_cu * 16;
    if (gx > cap) gx = cap;
    if (gy > 65535) gy = 65535;
    if (gx * gy > cap) gy = std::max<long>(1, cap / gx);
    const dim3 g((unsigned)gx, (unsigned)gy);
    const char *s = (const char *)src;
    char *d = (char *)dst;
    switch (w) {
    case 16: hipLaunchKernelGGL(copy2d_kernel<16>, g, dim3(256), 0, rt->stream, s, d, (long)rows, items, (long)src_pitch, (long)dst_pitch); break;
    case 8: hipLaunchKernelGGL(copy2d_kernel<8>, g, dim3(256), 0, rt->stream, s, d, (long)rows, items, (long)src_pitch, (long)dst_pitch); break;
    case 4: hipLaunchKernelGGL(copy2d_kernel<4>, g, dim3(256), 0, rt->stream, s, d, (long)rows, items, (long)src_pitch, (long)dst_pitch); break;
    case 2: hipLaunchKernelGGL(copy2d_kernel<2>, g, dim3(256), 0, rt->stream, s, d, (long)rows, items, (long)src_pitch, (long)dst_pitch); break;
    default: hipLaunchKernelGGL(copy2d_kernel<1>, g, dim3(256), 0, rt->stream, s, d, (long)rows, items, (long)src_pitch, (long)dst_pitch); break;
    }
    IROCM_LAUNCH_CHECK("strided_copy");
    return INFINI_ROCM_OK;
}

int infini_rocm_strided_copy_multi(infiniRocmRuntime_t rt, int count, const void *const *srcs, void *const *dsts, int64_t rows,
                                   const int64_t *row_bytes, const int64_t *src_pitch, const int64_t *dst_pitch) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(count >= 0 && rows >= 0, "strided_copy_multi: negative extent");
    if (count == 0 || rows == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(srcs && dsts && row_bytes && src_pitch && dst_pitch, "strided_copy_multi: NULL argument");
    for (int i = 0; i < count;) { // (more than kCopySegs non-empty segments: one launch per kCopySegs; `i` runs across the launches)
        MultiCopyArgs a;
        int n = 0, w = 16;
        long max_items = 0;
        for (; i < count && n < kCopySegs; ++i) {
            IROCM_CHECK_ARG(row_bytes[i] >= 0, "strided_copy_multi: negative extent");
            if (row_bytes[i] == 0)
                continue; // (an empty Concat input: the reference accepts it, test_cuda_concat.cc:160-190)
            IROCM_CHECK_ARG(srcs[i] && dsts[i], "strided_copy_multi: NULL tensor");
            while (w > 1 && ((row_bytes[i] % w) || (src_pitch[i] % w) || (dst_pitch[i] % w) || ((uintptr_t)srcs[i] % w) || ((uintptr_t)dsts[i] % w)))
                w >>= 1;
            a.src[n] = (const char *)srcs[i]; a.dst[n] = (char *)dsts[i];
            a.row_items[n] = row_bytes[i]; a.src_pitch[n] = src_pitch[i]; a.dst_pitch[n] = dst_pitch[i];
            ++n;
        }
        if (n == 0)
            continue;
        for (int i = 0; i < n; ++i) {
            a.row_items[i] /= w;
            max_items = std::max(max_items, a.row_items[i]);
        }
        a.rows = rows;
        const long cap = std::max<long>(1, (long)rt->num_cu * 16 / n);
        long gx = ceil_div(max_items, 256);
        if (gx > cap) gx = cap;
        const int u = rows >= 4 * std::max<long>(1, cap / gx) * 4 ? 4 : 1; // rows per trip (see the kernel)
        long gy = ceil_div(rows, u);
        if (gy > 65535) gy = 65535;
        if (gx * gy > cap) gy = std::max<long>(1, cap / gx);
        const dim3 g((unsigned)gx, (unsigned)gy, (unsigned)n);
#define IROCM_COPYM(B_)                                                                            \
    do {                                                                                           \
        if (u == 4) hipLaunchKernelGGL((copy2d_multi_kernel<B_, 4>), g, dim3(256), 0, rt->stream, a); \
        else hipLaunchKernelGGL((copy2d_multi_kernel<B_, 1>), g, dim3(256), 0, rt->stream, a);     \
    } while (0)
        switch (w) {
        case 16: IROCM_COPYM(16); break;
        case 8: IROCM_COPYM(8); break;
        case 4: IROCM_COPYM(4); break;
        case 2: IROCM_COPYM(2); break;
        default: IROCM_COPYM(1); break;
        }
#undef IROCM_COPYM
        IROCM_LAUNCH_CHECK("strided_copy_multi");
    }
    return INFINI_ROCM_OK;
}

} // extern "C"
