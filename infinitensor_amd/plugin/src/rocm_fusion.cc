// Launch-time operator fusion for Device::ROCM: the reference executes one kernel per operator
// (cuda_runtime.cc:136-170); on MI355X the element-wise tails of a convolution are pure HBM traffic, so the
// runtime folds them into the producing kernel's epilogue when that cannot be observed:
//
//   Conv -> Add(per-channel bias [1,F,1,1] / [F,1,1]) [-> Relu]   =>  conv2d(bias, act)        (onnx.py:159-190
//   Conv -> Relu                                                   =>  conv2d(act)              emits these chains)
//   Add  -> Relu                                                   =>  binary(ADD_RELU)         (residual join)
//
// Conditions (checked every launch, nothing is cached across graph mutations):
//   * the ops are CONSECUTIVE in the graph's operator order and each intermediate tensor has exactly one
//     consumer (so it is not a graph output and nobody else reads it);
//   * the buffer of the final tensor does not overlap an input of the fused kernel — the memory planner reuses
//     dead tensors' storage, and the fused kernel writes the final tensor earlier than the unfused chain would
//     (exception: Add -> Relu exactly in place over a same-extent input, which is index-wise safe);
//   * same dtype throughout. Intermediate tensors are simply not materialised.
// Numerics: fp32 is bit-identical (same operations in the same order); f16 / bf16 round once instead of after
// every op (a result at least as close to the exact value); Add -> Relu is bit-identical in every type.
// INFINI_ROCM_FUSION=0 or RocmRuntimeObj::setFusion(false) restores one kernel per operator.
#include "operators/conv.h"
#include "operators/element_wise.h"
#include "operators/unary.h"
#include "rocm/rocm_runtime.h"

namespace infini {

namespace {
bool overlaps(const Tensor &a, const Tensor &b) {
    const auto pa = reinterpret_cast<uintptr_t>(a->getRawDataPtr<void *>());
    const auto pb = reinterpret_cast<uintptr_t>(b->getRawDataPtr<void *>());
    return pa < pb + b->getBytes() && pb < pa + a->getBytes();
}
// `t` is consumed only by `next`, which directly follows in the operator order
bool soleConsumerIs(const Tensor &t, const Operator &next) {
    const auto targets = t->getTargets();
    return targets.size() == 1 && targets[0] == next;
}
bool isChannelBias(const Tensor &b, int f) {
    const auto &d = b->getDims();
    if (d.size() == 4)
        return d[0] == 1 && d[1] == f && d[2] == 1 && d[3] == 1;
    if (d.size() == 3)
        return d[0] == f && d[1] == 1 && d[2] == 1;
    return false;
}
std::vector<int64_t> strides64(const Shape &shape, const Shape &outShape) {
    const int r = shape.size(), ro = outShape.size();
    std::vector<int64_t> dense(r), out(ro, 0);
    int64_t p = 1;
    for (int i = r - 1; i >= 0; --i) {
        dense[i] = p;
        p *= shape[i];
    }
    for (int i = 0; i < ro; ++i) {
        const int j = i - (ro - r);
        if (j >= 0 && shape[j] != 1)
            out[i] = dense[j];
    }
    return out;
}
} // namespace

size_t RocmRuntimeObj::tryLaunchFused(const OpVec &ops, size_t i) const {
    const Operator &op = ops[i];
    const auto type = op->getOpType();
    if (type == OpType::Conv) {
        auto conv = as<ConvObj>(op);
        const Tensor x = conv->getInputs(0), w = conv->getInputs(1);
        Tensor last = conv->getOutput(), bias = nullptr;
        const int f = last->getDims()[1];
        size_t used = 1;
        int act = 0;
        if (i + used < ops.size() && ops[i + used]->getOpType() == OpType::Add && soleConsumerIs(last, ops[i + used])) {
            const Operator &add = ops[i + used];
            const Tensor a0 = add->getInputs(0), a1 = add->getInputs(1);
            const Tensor other = a0 == last ? a1 : a0;
            if (other != last && isChannelBias(other, f) && other->getDType() == last->getDType() &&
                add->getOutput()->getDims() == last->getDims()) {
                bias = other;
                last = add->getOutput();
                ++used;
            }
        }
        if (i + used < ops.size() && ops[i + used]->getOpType() == OpType::Relu && soleConsumerIs(last, ops[i + used])) {
            act = 1;
            last = ops[i + used]->getOutput();
            ++used;
        }
        if (used == 1)
            return 0;
        if (!(last->getDType() == x->getDType()) || overlaps(last, x) || overlaps(last, w) || (bias && overlaps(last, bias)))
            return 0;
        const auto [n, c, h, wd, ff, r, s] = conv->getNCHWFRS();
        const auto [ph, pw, sh, sw, dh, dw] = conv->getPadStrideDilation();
        ROCM_CALL(infini_rocm_conv2d(rt, x->getDTypeIndex(), x->getRawDataPtr<void *>(), w->getRawDataPtr<void *>(),
                                     bias ? bias->getRawDataPtr<void *>() : nullptr, last->getRawDataPtr<void *>(), n, c, h,
                                     wd, ff, r, s, ph, pw, sh, sw, dh, dw, conv->getNumGroups(), act));
        return used;
    }
    if (type == OpType::Add && i + 1 < ops.size() && ops[i + 1]->getOpType() == OpType::Relu &&
        soleConsumerIs(op->getOutput(), ops[i + 1])) {
        const Tensor a = op->getInputs(0), b = op->getInputs(1), out = ops[i + 1]->getOutput();
        const auto &od = op->getOutput()->getDims();
        // element-wise with identical extents may run exactly in place (the planner likes to give the Relu output the
        // storage of a dead Add input); any other overlap is a hazard
        auto hazard = [&](const Tensor &t) {
            const bool inPlace = t->getRawDataPtr<void *>() == out->getRawDataPtr<void *>() && t->getDims() == od;
            return overlaps(out, t) && !inPlace;
        };
        if (!(a->getDType() == b->getDType()) || hazard(a) || hazard(b))
            return 0;
        const auto shape = std::vector<int64_t>(od.begin(), od.end());
        const auto sa = strides64(a->getDims(), od), sb = strides64(b->getDims(), od);
        ROCM_CALL(infini_rocm_binary(rt, INFINI_BIN_ADD_RELU, a->getDTypeIndex(), a->getRawDataPtr<void *>(),
                                     b->getRawDataPtr<void *>(), out->getRawDataPtr<void *>(), (int)shape.size(),
                                     shape.data(), sa.data(), sb.data()));
        return 2;
    }
    return 0;
}

} // namespace infini
