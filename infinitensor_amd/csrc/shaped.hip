// "From-shape" entry points: the shape -> stride / extent glue of the operators whose C ABI speaks strides, written ONCE below the
// ABI. Until round 5 this arithmetic existed twice — in plugin/src/rocm_kernels.cc (the reference executor's kernels) and in
// infinitensor_amd/ops.py (the ctypes mirror the C-ABI tests and bench.py use) — so each suite exercised its own copy and a divergence
// between the two was invisible to both. Both callers now hand over the operands' SHAPES as the reference's operators hold them
// (TensorObj::getDims()) and this file derives what the stride-level entry points want:
//   broadcast strides        reference: infer_broadcast / the kernels' own index arithmetic, src/utils/operator_utils.cc:6-32,
//                            src/kernels/cuda/element_wise.cu:9-60 (4-D padded shapes there, any rank <= 8 here)
//   MatMul batch / bias      src/kernels/cuda/matmul.cc:86-137 (batch broadcast by zero stride; bias expanded to [.., m, n])
//   Concat / Split segments  src/kernels/cuda/split_concat.cc:10-62 (one segment per input / output along the axis)
//   Pad                      include/operators/pad.h (pads = begin_0 .. begin_{r-1}, end_0 .. end_{r-1}; constant 0)
//   Gather extents           include/operators/gather.h:27-49
// Pure host code: no kernel lives here. Errors: INVALID_ARGUMENT with a message (infini_rocm_last_error).
#include "common.h"

#include <vector>

using namespace irocm;

namespace {

int64_t prod_range(const int64_t *s, int from, int to) {
    int64_t p = 1;
    for (int i = from; i < to; ++i)
        p *= s[i];
    return p;
}

// element strides of a dense tensor of `shape` viewed in `out_shape` (0 where broadcast); false when not broadcastable
bool bcast(int rank, const int64_t *shape, int out_rank, const int64_t *out_shape, int64_t *strides) {
    if (rank > out_rank)
        return false;
    int64_t p = 1;
    for (int i = out_rank - 1; i >= 0; --i) {
        const int j = i - (out_rank - rank);
        if (j < 0) {
            strides[i] = 0;
            continue;
        }
        if (shape[j] != 1 && shape[j] != out_shape[i])
            return false;
        strides[i] = shape[j] != 1 ? p : 0;
        p *= shape[j];
    }
    return true;
}

} // namespace

extern "C" int infini_rocm_broadcast_strides(int rank, const int64_t *shape, int out_rank, const int64_t *out_shape, int64_t *strides) {
    IROCM_CHECK_ARG(rank >= 0 && out_rank >= 0 && out_rank <= INFINI_ROCM_MAX_DIMS && (rank == 0 || shape) && (out_rank == 0 || (out_shape && strides)),
                    "broadcast_strides: bad rank %d -> %d (at most %d dims)", rank, out_rank, INFINI_ROCM_MAX_DIMS);
    IROCM_CHECK_ARG(bcast(rank, shape, out_rank, out_shape, strides), "broadcast_strides: a rank-%d shape does not broadcast to the rank-%d output",
                    rank, out_rank);
    return INFINI_ROCM_OK;
}

extern "C" int infini_rocm_binary_shaped(infiniRocmRuntime_t rt, int op, int dtype, const void *a, int a_rank, const int64_t *a_shape,
                                         const void *b, int b_rank, const int64_t *b_shape, void *out, int out_rank, const int64_t *out_shape) {
    int64_t sa[INFINI_ROCM_MAX_DIMS], sb[INFINI_ROCM_MAX_DIMS];
    int st = infini_rocm_broadcast_strides(a_rank, a_shape, out_rank, out_shape, sa);
    if (st == INFINI_ROCM_OK)
        st = infini_rocm_broadcast_strides(b_rank, b_shape, out_rank, out_shape, sb);
    if (st != INFINI_ROCM_OK)
        return st;
    return infini_rocm_binary(rt, op, dtype, a, b, out, out_rank, out_shape, sa, sb);
}

extern "C" int infini_rocm_where_shaped(infiniRocmRuntime_t rt, int dtype, int cond_dtype, const void *x, int x_rank, const int64_t *x_shape,
                                        const void *y, int y_rank, const int64_t *y_shape, const void *cond, int c_rank, const int64_t *c_shape,
                                        void *out, int out_rank, const int64_t *out_shape) {
    int64_t sx[INFINI_ROCM_MAX_DIMS], sy[INFINI_ROCM_MAX_DIMS], sc[INFINI_ROCM_MAX_DIMS];
    int st = infini_rocm_broadcast_strides(x_rank, x_shape, out_rank, out_shape, sx);
    if (st == INFINI_ROCM_OK)
        st = infini_rocm_broadcast_strides(y_rank, y_shape, out_rank, out_shape, sy);
    if (st == INFINI_ROCM_OK)
        st = infini_rocm_broadcast_strides(c_rank, c_shape, out_rank, out_shape, sc);
    if (st != INFINI_ROCM_OK)
        return st;
    return infini_rocm_where_ex(rt, dtype, cond_dtype, x, y, cond, out, out_rank, out_shape, sx, sy, sc);
}

extern "C" int infini_rocm_expand_shaped(infiniRocmRuntime_t rt, int dtype, const void *x, int x_rank, const int64_t *x_shape, void *out,
                                         int out_rank, const int64_t *out_shape) {
    int64_t sx[INFINI_ROCM_MAX_DIMS];
    const int st = infini_rocm_broadcast_strides(x_rank, x_shape, out_rank, out_shape, sx);
    if (st != INFINI_ROCM_OK)
        return st;
    return infini_rocm_expand(rt, dtype, x, out, out_rank, out_shape, sx);
}

extern "C" int infini_rocm_matmul_plan(int a_rank, const int64_t *a_shape, int b_rank, const int64_t *b_shape, int bias_rank,
                                       const int64_t *bias_shape, int trans_a, int trans_b, int64_t *plan) {
    IROCM_CHECK_ARG(a_rank >= 2 && b_rank >= 2 && a_shape && b_shape && plan && a_rank <= INFINI_ROCM_MAX_DIMS && b_rank <= INFINI_ROCM_MAX_DIMS,
                    "matmul: operands of rank %d and %d (2 .. %d)", a_rank, b_rank, INFINI_ROCM_MAX_DIMS);
    const int64_t m = trans_a ? a_shape[a_rank - 1] : a_shape[a_rank - 2], ka = trans_a ? a_shape[a_rank - 2] : a_shape[a_rank - 1];
    const int64_t n = trans_b ? b_shape[b_rank - 2] : b_shape[b_rank - 1], kb = trans_b ? b_shape[b_rank - 1] : b_shape[b_rank - 2];
    IROCM_CHECK_ARG(ka == kb, "matmul: K of A is %lld, K of B is %lld", (long long)ka, (long long)kb); // reference: IT_ASSERT(kA == kB)
    // batch = broadcast of the leading dims (matmul.cc:26-49); an operand takes part with its full batch or with batch 1 (zero stride)
    const int ra = a_rank - 2, rb = b_rank - 2, rbatch = ra > rb ? ra : rb;
    int64_t batch = 1;
    for (int i = 0; i < rbatch; ++i) {
        const int64_t da = i >= rbatch - ra ? a_shape[i - (rbatch - ra)] : 1, db = i >= rbatch - rb ? b_shape[i - (rbatch - rb)] : 1;
        IROCM_CHECK_ARG(da == db || da == 1 || db == 1, "matmul: batch dims %lld and %lld do not broadcast", (long long)da, (long long)db);
        batch *= (da == 0 || db == 0) ? 0 : (da > db ? da : db);
    }
    const int64_t ba = prod_range(a_shape, 0, ra), bb = prod_range(b_shape, 0, rb);
    IROCM_CHECK_ARG((ba == 1 || ba == batch) && (bb == 1 || bb == batch),
                    "matmul: only full or size-1 batch broadcast is supported (reference matmul.cc:124-137): %lld and %lld of %lld",
                    (long long)ba, (long long)bb, (long long)batch);
    plan[0] = batch; plan[1] = m; plan[2] = n; plan[3] = ka;
    plan[4] = (ba == 1 && batch > 1) ? 0 : m * ka;
    plan[5] = (bb == 1 && batch > 1) ? 0 : n * ka;
    plan[6] = plan[7] = plan[8] = 0;
    if (bias_rank >= 0 && bias_shape) { // the bias broadcast to [batch dims .., m, n] (matmul.cc:86-118), its batch dims as ONE stride
        const int ro = rbatch + 2;
        IROCM_CHECK_ARG(bias_rank <= ro, "matmul: bias of rank %d against an output of rank %d", bias_rank, ro);
        int64_t oshape[INFINI_ROCM_MAX_DIMS + 2], st[INFINI_ROCM_MAX_DIMS + 2];
        for (int i = 0; i < rbatch; ++i) {
            const int64_t da = i >= rbatch - ra ? a_shape[i - (rbatch - ra)] : 1, db = i >= rbatch - rb ? b_shape[i - (rbatch - rb)] : 1;
            oshape[i] = da > db ? da : db;
        }
        oshape[ro - 2] = m;
        oshape[ro - 1] = n;
        IROCM_CHECK_ARG(bcast(bias_rank, bias_shape, ro, oshape, st), "matmul: the bias does not broadcast to the output");
        plan[7] = st[ro - 2];
        plan[8] = st[ro - 1];
        bool lead = false;
        for (int i = 0; i < rbatch; ++i)
            lead = lead || (st[i] != 0 && oshape[i] != 1);
        if (lead) {
            IROCM_CHECK_ARG(bias_rank >= 2 && prod_range(bias_shape, 0, bias_rank - 2) == batch, "matmul: partially broadcast bias batch is not supported");
            plan[6] = bias_shape[bias_rank - 2] * bias_shape[bias_rank - 1];
        }
    }
    return INFINI_ROCM_OK;
}

extern "C" int infini_rocm_matmul_shaped(infiniRocmRuntime_t rt, int dtype, const void *a, int a_rank, const int64_t *a_shape, const void *b,
                                         int b_rank, const int64_t *b_shape, const void *bias, int bias_rank, const int64_t *bias_shape, void *out,
                                         int trans_a, int trans_b, int act, int64_t seq, int64_t head_dim) {
    int64_t p[9];
    const int st = infini_rocm_matmul_plan(a_rank, a_shape, b_rank, b_shape, bias ? bias_rank : -1, bias ? bias_shape : nullptr, trans_a, trans_b, p);
    if (st != INFINI_ROCM_OK)
        return st;
    return infini_rocm_matmul_headsplit(rt, dtype, a, b, bias, out, p[0], p[1], p[2], p[3], trans_a, trans_b, p[4], p[5], p[6], p[7], p[8], act,
                                        seq, head_dim);
}

// Concat: input i holds axis_extents[i] slices of the output's axis; empty inputs (the reference accepts them, test_cuda_concat.cc:160-190)
// are skipped by the copy. Split is the same walk with the roles of the pitches swapped.
static int concat_split(infiniRocmRuntime_t rt, bool is_split, int elem_size, int count, const void *const *parts, const int64_t *axis_extents,
                        const void *whole, int rank, const int64_t *shape, int axis) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(count >= 1 && parts && axis_extents && shape && rank >= 1 && rank <= INFINI_ROCM_MAX_DIMS && axis >= 0 && axis < rank && elem_size > 0,
                    "%s: bad arguments (count %d, rank %d, axis %d)", is_split ? "split" : "concat", count, rank, axis);
    int64_t sum = 0;
    for (int i = 0; i < count; ++i) {
        IROCM_CHECK_ARG(axis_extents[i] >= 0, "%s: negative extent", is_split ? "split" : "concat");
        sum += axis_extents[i];
    }
    IROCM_CHECK_ARG(sum == shape[axis], "%s: the parts' extents add up to %lld, the axis has %lld", is_split ? "split" : "concat", (long long)sum,
                    (long long)shape[axis]);
    const int64_t outer = prod_range(shape, 0, axis), inner_bytes = prod_range(shape, axis + 1, rank) * elem_size;
    const int64_t pitch = shape[axis] * inner_bytes;
    std::vector<const void *> srcs(count);
    std::vector<void *> dsts(count);
    std::vector<int64_t> rbs(count), pitches(count, pitch);
    int64_t off = 0;
    for (int i = 0; i < count; ++i) {
        const int64_t rb = axis_extents[i] * inner_bytes;
        rbs[i] = rb;
        if (is_split) {
            srcs[i] = (const char *)whole + off;
            dsts[i] = const_cast<void *>(parts[i]);
        } else {
            srcs[i] = rb ? parts[i] : nullptr;
            dsts[i] = (char *)const_cast<void *>(whole) + off;
        }
        off += rb;
    }
    return is_split ? infini_rocm_strided_copy_multi(rt, count, srcs.data(), dsts.data(), outer, rbs.data(), pitches.data(), rbs.data())
                    : infini_rocm_strided_copy_multi(rt, count, srcs.data(), dsts.data(), outer, rbs.data(), rbs.data(), pitches.data());
}

extern "C" int infini_rocm_concat_shaped(infiniRocmRuntime_t rt, int elem_size, int count, const void *const *inputs, const int64_t *axis_extents,
                                         void *out, int out_rank, const int64_t *out_shape, int axis) {
    return concat_split(rt, false, elem_size, count, inputs, axis_extents, out, out_rank, out_shape, axis);
}

extern "C" int infini_rocm_split_shaped(infiniRocmRuntime_t rt, int elem_size, int count, void *const *outputs, const int64_t *axis_extents,
                                        const void *in, int in_rank, const int64_t *in_shape, int axis) {
    return concat_split(rt, true, elem_size, count, (const void *const *)outputs, axis_extents, in, in_rank, in_shape, axis);
}

extern "C" int infini_rocm_pad_shaped(infiniRocmRuntime_t rt, int dtype, const void *x, void *y, int rank, const int64_t *in_shape,
                                      const int64_t *pads) {
    IROCM_CHECK_ARG(rank >= 1 && rank <= INFINI_ROCM_MAX_DIMS && in_shape && pads, "pad: bad rank %d", rank);
    int64_t oshape[INFINI_ROCM_MAX_DIMS], starts[INFINI_ROCM_MAX_DIMS];
    for (int d = 0; d < rank; ++d) {
        IROCM_CHECK_ARG(pads[d] >= 0 && pads[d + rank] >= 0, "pad: negative pad on dim %d (the reference operator carries none)", d);
        oshape[d] = in_shape[d] + pads[d] + pads[d + rank];
        starts[d] = -pads[d];
    }
    return infini_rocm_pad_slice(rt, dtype, x, y, rank, in_shape, oshape, starts, nullptr, 0);
}

extern "C" int infini_rocm_gather_shaped(infiniRocmRuntime_t rt, int dtype, int index_dtype, const void *data, int data_rank,
                                         const int64_t *data_shape, const void *indices, int64_t n_indices, void *out, int axis) {
    IROCM_CHECK_ARG(data_rank >= 1 && data_rank <= INFINI_ROCM_MAX_DIMS && data_shape && axis >= 0 && axis < data_rank, "gather: axis %d of rank %d",
                    axis, data_rank);
    return infini_rocm_gather(rt, dtype, index_dtype, data, indices, out, prod_range(data_shape, 0, axis), data_shape[axis], n_indices,
                              prod_range(data_shape, axis + 1, data_rank));
}
