// Launch-time operator fusion for Device::ROCM: the reference executes one kernel per operator
// (cuda_runtime.cc:136-170); on MI355X the element-wise tails of a convolution are pure HBM traffic, so the
// runtime folds them into the producing kernel's epilogue when that cannot be observed:
//
//   Conv -> Add(per-channel bias [1,F,1,1] / [F,1,1]) [-> Relu]   =>  conv2d(bias, act)        (onnx.py:159-190
//   Conv -> Relu                                                   =>  conv2d(act)              emits these chains)
//   Conv -> Add(bias) -> Add(identity) [-> Relu]                   =>  conv2d_res(bias, residual, act)   [f16 / bf16,
//        even output planes: the residual moves in the LDS-staged epilogue's 128-byte rows; INFINI_ROCM_FUSE_RES=0|1]
//   Add  -> Relu                                                   =>  binary(ADD_RELU)         (residual join)
//   Relu -> MaxPool                                                =>  pool2d_relu              (ResNet stem)
//   Add  -> LayerNormalization(last axis) | RMSNorm                =>  add_norm                 (transformer residual)
//   Add(bias) -> Add(identity) [-> Relu]                           =>  bias_residual            (when the conv could not
//        take the bias: its input's storage was recycled for the output)
//   MatMul(+bias) -> Reshape [B,S,H,D] -> Transpose(0,2,1,3)        =>  matmul_headsplit (the q / k / v head split in
//        the GEMM epilogue; f16 / bf16)
//   MatMul(+bias) -> Gelu (f16 / bf16)                              =>  matmul(act = 5): Gelu in the GEMM epilogue
//   MatMul | Transpose | element-wise | Softmax | LayerNorm | Gather -> Reshape-family copy
//                                                                  =>  the producer writes into the copy's output
//   MatMul(Q, K^T) [-> Div|Mul(scalar)] [-> Add(mask)] -> Softmax(last axis) -> MatMul(P, V)
//                                                                  =>  attention (csrc/attention.hip): the score
//        matrix is never written; Q, K, V rank-4 [b, h, S, D] with D in {64, 128}, f16 / bf16, mask [b|1, 1, 1, Sk];
//        a following Transpose(0,2,1,3) -> Reshape (the head merge) is folded into the kernel's store
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
#include "core/perf_engine.h"
#include "operators/conv.h"
#include "rocm/rocm_perf.h"
#include <cstdio>
#include <cstdlib>
#include <string>
#include "operators/element_wise.h"
#include "operators/layer_norm.h"
#include "operators/rms_norm.h"
#include "operators/matmul.h"
#include "operators/pooling.h"
#include "operators/softmax.h"
#include "operators/transpose.h"
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
// `b` is a per-channel bias of a tensor with dims `d` ([N, C, ...]): [1, C, 1, ...] or [C, 1, ...]
bool isChannelBiasOf(const Tensor &b, const Shape &d) {
    const auto &bd = b->getDims();
    if ((int64_t)b->size() != d[1])
        return false;
    if (bd.size() == d.size()) {
        for (size_t i = 0; i < bd.size(); ++i)
            if (bd[i] != (i == 1 ? d[1] : 1))
                return false;
        return true;
    }
    if (bd.size() + 1 == d.size()) {
        for (size_t i = 0; i < bd.size(); ++i)
            if (bd[i] != (i == 0 ? d[1] : 1))
                return false;
        return true;
    }
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

namespace {
struct OutputRedirect { // RAII: the redirection never outlives one launch
    OutputRedirect(const TensorObj *t, void *p, int seq = 0, int headDim = 0, int act = 0) {
        RocmRuntimeObj::redirectTensor = t;
        RocmRuntimeObj::redirectPtr = p;
        RocmRuntimeObj::redirectSeq = seq;
        RocmRuntimeObj::redirectHeadDim = headDim;
        RocmRuntimeObj::redirectAct = act;
    }
    ~OutputRedirect() {
        RocmRuntimeObj::redirectTensor = nullptr;
        RocmRuntimeObj::redirectPtr = nullptr;
        RocmRuntimeObj::redirectSeq = RocmRuntimeObj::redirectHeadDim = RocmRuntimeObj::redirectAct = 0;
    }
};
} // namespace

void RocmRuntimeObj::launchWithInputRedirect(const Operator &op, const TensorObj *t, void *ptr) const {
    OutputRedirect feed(t, ptr); // P(t) resolves to ptr for the duration of this one launch (rocm_kernels.cc)
    launchOne(op);
}

size_t RocmRuntimeObj::tryLaunchFusedAttention(const OpVec &ops, size_t i) const {
    auto mm1 = as<MatmulObj>(ops[i]);
    if (mm1->getTransA() || !mm1->getTransB() || mm1->getBias() || mm1->getAct() != ActType::None)
        return 0;
    const Tensor q = mm1->getInputs(0), k = mm1->getInputs(1);
    const auto &qd = q->getDims(), &kd = k->getDims();
    const int dt = q->getDTypeIndex();
    if (qd.size() != 4 || kd.size() != 4 || qd[0] != kd[0] || qd[1] != kd[1] || qd[3] != kd[3] || (qd[3] != 64 && qd[3] != 128) ||
        (dt != INFINI_DT_F16 && dt != INFINI_DT_BF16) || !(k->getDType() == q->getDType()))
        return 0;
    const int b = qd[0], h = qd[1], sq = qd[2], sk = kd[2], d = qd[3];
    // limits of infini_rocm_attention_ex (attention.hip): outside them the chain simply runs unfused
    if ((int64_t)b * h >= 65536 || h >= 65536 || sk <= 0 || sq <= 0 ||
        ((((uintptr_t)q->getRawDataPtr<void *>()) | ((uintptr_t)k->getRawDataPtr<void *>())) & 15) != 0)
        return 0;
    Tensor cur = mm1->getOutput(), scale = nullptr, mask = nullptr;
    bool mask2d = false;
    size_t j = i + 1;
    bool isDiv = false;
    auto next = [&](OpType t) { return j < ops.size() && ops[j]->getOpType() == t && soleConsumerIs(cur, ops[j]); };
    if (next(OpType::Div) || next(OpType::Mul)) {
        const Tensor a0 = ops[j]->getInputs(0), a1 = ops[j]->getInputs(1);
        isDiv = ops[j]->getOpType() == OpType::Div;
        const Tensor other = a0 == cur ? a1 : a0;
        if (other == cur || other->size() != 1 || !(other->getDType() == q->getDType()) || (isDiv && a0 != cur))
            return 0;
        scale = other;
        cur = ops[j++]->getOutput();
    }
    if (next(OpType::Add)) {
        const Tensor a0 = ops[j]->getInputs(0), a1 = ops[j]->getInputs(1);
        const Tensor other = a0 == cur ? a1 : a0;
        const auto &md = other->getDims();
        // key mask [b|1, 1, 1, Sk] (BERT padding) or a full additive mask [b|1, h|1, Sq, Sk] with the same grouping rule:
        // heads may only broadcast when the batch does too or both are explicit ([1,1], [b,1], [b,h])
        const bool keyMask = md.size() == 4 && md[1] == 1 && md[2] == 1;
        const bool fullMask = md.size() == 4 && md[2] == sq && sq > 1 && (md[1] == 1 || (md[1] == h && md[0] == b));
        if (other == cur || md.size() != 4 || (md[0] != b && md[0] != 1) || !(keyMask || fullMask) || md[3] != sk ||
            !(other->getDType() == q->getDType()))
            return 0;
        mask = other;
        mask2d = !keyMask;
        cur = ops[j++]->getOutput();
    }
    if (!next(OpType::Softmax) || as<SoftmaxObj>(ops[j])->getAxis() != 3)
        return 0;
    cur = ops[j++]->getOutput();
    if (!next(OpType::MatMul))
        return 0;
    auto mm2 = as<MatmulObj>(ops[j]);
    const Tensor v = mm2->getInputs(1), out = mm2->getOutput();
    if (mm2->getInputs(0) != cur || mm2->getTransA() || mm2->getTransB() || mm2->getBias() || mm2->getAct() != ActType::None ||
        v->getDims() != kd || !(v->getDType() == q->getDType()) || (((uintptr_t)v->getRawDataPtr<void *>()) & 15) != 0 ||
        (((uintptr_t)out->getRawDataPtr<void *>()) & 7) != 0)
        return 0;
    // Head merge: ctx [b, h, Sq, D] -> Transpose(0, 2, 1, 3) -> Reshape [b, Sq, h * D] (what every exported transformer
    // does before the output projection) is folded into the kernel's store (infini_rocm_attention_ex).
    Tensor dstT = out;
    int64_t heads = 0;
    size_t last = j;
    static const bool mergeOn = !(std::getenv("INFINI_ROCM_FUSE_HEADMERGE") && std::atoi(std::getenv("INFINI_ROCM_FUSE_HEADMERGE")) == 0);
    if (mergeOn && j + 2 < ops.size() && ops[j + 1]->getOpType() == OpType::Transpose && ops[j + 2]->getOpType() == OpType::Reshape &&
        soleConsumerIs(out, ops[j + 1]) && soleConsumerIs(ops[j + 1]->getOutput(), ops[j + 2]) &&
        ops[j + 2]->getInputs(0) == ops[j + 1]->getOutput()) {
        const auto perm = as<TransposeObj>(ops[j + 1])->getPermute();
        const Tensor r = ops[j + 2]->getOutput();
        if (perm.size() == 4 && perm[0] == 0 && perm[1] == 2 && perm[2] == 1 && perm[3] == 3 && r->getBytes() == out->getBytes() &&
            r->getDType() == out->getDType()) {
            dstT = r;
            heads = h;
            last = j + 2;
        }
    }
    // O may sit exactly on Q (the planner likes to: Q is dead after the first MatMul and has O's size): a workgroup
    // loads its query rows before the key sweep and writes the same rows of O after it (plain layout only). K / V are
    // read by everyone. Any other overlap (K and V die after their MatMul too, and have O's size) is bridged through the
    // workspace: O is [Sq, D] per head, the copy is small next to the score traffic the fusion removes.
    const bool onQ = heads == 0 && dstT->getRawDataPtr<void *>() == q->getRawDataPtr<void *>() && dstT->getDims() == qd;
    const bool hazard = (overlaps(dstT, q) && !onQ) || overlaps(dstT, k) || overlaps(dstT, v) ||
                        (mask && overlaps(dstT, mask)) || (scale && overlaps(dstT, scale));
    void *dst = dstT->getRawDataPtr<void *>();
    if (hazard)
        dst = getWorkspace(dstT->getBytes());
    // pairs (batch, head) served by one mask slab: [1,1,..] all of them, [b,1,..] the heads of a batch, [b,h,..] one
    const int64_t group = !mask ? 1 : ((mask->getDims()[1] == h && h > 1) ? 1 : (mask->getDims()[0] == 1 ? (int64_t)b * h : h));
    ROCM_CALL(infini_rocm_attention_ex(rt, dt, q->getRawDataPtr<void *>(), k->getRawDataPtr<void *>(),
                                              v->getRawDataPtr<void *>(), mask ? mask->getRawDataPtr<void *>() : nullptr, dst,
                                              (int64_t)b * h, sq, sk, d, group, scale ? scale->getRawDataPtr<void *>() : nullptr,
                                       isDiv ? 1 : 0, 1.0f, 0, heads, mask2d ? 1 : 0));
    if (hazard) {
        // The bridged result usually feeds exactly one operator, the output projection, which is next in the list: let
        // that MatMul read its A operand straight from the workspace (input redirect through P(), rocm_kernels.cc) instead
        // of copying 25 MB per BERT layer to a tensor nobody else reads. Only a plain MatMul that cannot itself take the
        // workspace (split-K partial planes) and that no other launch-time rule would pick up.
        static const bool feedOn = !(std::getenv("INFINI_ROCM_FEED_NEXT") && std::atoi(std::getenv("INFINI_ROCM_FEED_NEXT")) == 0);
        const size_t nx = last + 1;
        if (feedOn && nx < ops.size() && ops[nx]->getOpType() == OpType::MatMul && soleConsumerIs(dstT, ops[nx])) {
            auto mmn = as<MatmulObj>(ops[nx]);
            const auto [nb, nm, nn, nk] = mmn->getBMNK();
            int mayWs = 1;
            ROCM_CALL(infini_rocm_matmul_may_use_workspace(rt, nb, nm, nn, &mayWs));
            bool onlyA = mmn->getInputs(0) == dstT;
            for (size_t q = 1; q < mmn->getInputs().size(); ++q)
                onlyA = onlyA && mmn->getInputs(q) != dstT;
            bool otherRule = false; // head split / Gelu / copy elision would want the single redirect slot themselves
            if (nx + 1 < ops.size()) {
                const auto t2 = ops[nx + 1]->getOpType();
                otherRule = t2 == OpType::Reshape || t2 == OpType::Flatten || t2 == OpType::Identity || t2 == OpType::Squeeze ||
                            t2 == OpType::Unsqueeze || t2 == OpType::Gelu;
            }
            if (!mayWs && onlyA && !otherRule && tunedVariant(ops[nx]) != 3 && !overlaps(mmn->getOutput(), dstT)) {
                OutputRedirect feed(dstT.get(), dst);
                launchOne(ops[nx]);
                return last + 2 - i;
            }
        }
        ROCM_CALL(infini_rocm_copy_inside(rt, dstT->getRawDataPtr<void *>(), dst, dstT->getBytes()));
    }
    return last + 1 - i;
}

int RocmRuntimeObj::tunedVariant(const Operator &op) const {
    auto key = PerfEngine::Key{KernelAttrs{device, op->getOpType().underlying()}, op->getOpPerfKey()};
    auto rec = std::dynamic_pointer_cast<RocmVariantPerfRecordObj>(PerfEngine::getInstance().getPerfData(key));
    return rec ? rec->variant : -1;
}

size_t RocmRuntimeObj::tryLaunchFused(const OpVec &ops, size_t i) const {
    if (const size_t used = tryLaunchFusedRules(ops, i))
        return used;
    if (const size_t used = tryLaunchHeadSplit(ops, i))
        return used;
    if (const size_t used = tryLaunchMatmulGelu(ops, i))
        return used;
    if (const size_t used = tryLaunchGroupedMatmul(ops, i))
        return used;
    if (const size_t used = tryLaunchRopeHeadSplit(ops, i))
        return used;
    return tryLaunchIntoReshape(ops, i);
}


// MatMul(+bias) [.., S, H*D] or [B*S, H*D] -> Reshape [B, S, H, D] -> Transpose(0, 2, 1, 3): the head split of a
// transformer's q / k / v projections (three launches and two extra passes over the activation in the reference) as ONE
// GEMM whose epilogue stores head-split (infini_rocm_matmul_headsplit). Same sums, same rounding: bit-identical.
//
// Two or three such chains in a row that read the SAME activations (q, k, v) become ONE grouped launch when their weights,
// biases and outputs happen to sit at a uniform spacing in memory (the reference's allocator hands out a layer's weights
// back to back, and the three head-split outputs are the same size): the group index is the GEMM's batch index with a zero
// A stride. One launch walks 3 x 256 tiles per CU-set instead of three launches of one tile per CU each paying the pipeline
// prologue and the launch gap (BERT-base: 3 x 25 us -> ~62 us per layer). Same kernels, same sums: bit-identical.
namespace {
struct HeadSplitChain {
    std::shared_ptr<MatmulObj> mm;
    Tensor out;
    long S = 0, D = 0, rows = 0; // rows = b * m: the GEMM's row count with the batch folded in
    int n = 0, k = 0;
};
bool matchHeadSplit(const OpVec &ops, size_t i, HeadSplitChain &h) {
    if (i + 2 >= ops.size() || ops[i]->getOpType() != OpType::MatMul || ops[i + 1]->getOpType() != OpType::Reshape ||
        ops[i + 2]->getOpType() != OpType::Transpose)
        return false;
    auto mm = as<MatmulObj>(ops[i]);
    auto tr = as<TransposeObj>(ops[i + 2]);
    const Tensor c = mm->getOutput(), r = ops[i + 1]->getOutput(), out = tr->getOutput();
    if (ops[i + 1]->getInputs(0) != c || tr->getInputs(0) != r || !soleConsumerIs(c, ops[i + 1]) || !soleConsumerIs(r, ops[i + 2]))
        return false;
    const auto &rd = r->getDims();
    const auto perm = tr->getPermute();
    if (rd.size() != 4 || perm.size() != 4 || perm[0] != 0 || perm[1] != 2 || perm[2] != 1 || perm[3] != 3)
        return false;
    const auto [b, m, n, k] = mm->getBMNK();
    const long B = rd[0], S = rd[1], Hh = rd[2], D = rd[3];
    // the MatMul's rows are (batch, position), its columns (head, channel): [b x m] == [B x S] row-wise, n == H * D
    if ((long)b * m != B * S || (long)n != Hh * D || m % S != 0 || D % 8 != 0 ||
        !(c->getDType() == out->getDType()) || c->getBytes() != out->getBytes())
        return false;
    for (const auto &in : mm->getInputs())
        if (overlaps(out, in))
            return false;
    h.mm = mm;
    h.out = out;
    h.S = S;
    h.D = D;
    h.rows = (long)b * m;
    h.n = n;
    h.k = k;
    return true;
}
} // namespace

size_t RocmRuntimeObj::tryLaunchHeadSplit(const OpVec &ops, size_t i) const {
    static const bool enabled = !(std::getenv("INFINI_ROCM_FUSE_HEADSPLIT") && std::atoi(std::getenv("INFINI_ROCM_FUSE_HEADSPLIT")) == 0);
    HeadSplitChain h0;
    if (!enabled || !matchHeadSplit(ops, i, h0))
        return 0;
    // ---- grouped q / k / v ------------------------------------------------------------------------------
    static const bool groupOn = !(std::getenv("INFINI_ROCM_GROUP_QKV") && std::atoi(std::getenv("INFINI_ROCM_GROUP_QKV")) == 0);
    const auto &mm0 = h0.mm;
    const Tensor a0 = mm0->getInputs(0), w0 = mm0->getInputs(1);
    const Tensor bias0 = mm0->numInputs() == 3 ? mm0->getInputs(2) : nullptr;
    const int dt = a0->getDTypeIndex();
    const bool groupable = groupOn && (dt == INFINI_DT_F16 || dt == INFINI_DT_BF16) && !mm0->getTransA() && !mm0->getTransB() &&
                           w0->getRank() == 2 && tunedVariant(ops[i]) < 0 &&
                           (!bias0 || (bias0->getRank() == 1 && (int)bias0->size() == h0.n));
    if (groupable) {
        std::vector<HeadSplitChain> g{h0};
        while (g.size() < 4) {
            HeadSplitChain hn;
            if (!matchHeadSplit(ops, i + 3 * g.size(), hn))
                break;
            const auto &mn = hn.mm;
            const Tensor wn = mn->getInputs(1), bn = mn->numInputs() == 3 ? mn->getInputs(2) : nullptr;
            if (mn->getInputs(0) != a0 || mn->getTransA() || mn->getTransB() || wn->getDims() != w0->getDims() ||
                !(wn->getDType() == w0->getDType()) || (bn != nullptr) != (bias0 != nullptr) ||
                (bn && bn->getDims() != bias0->getDims()) || hn.S != h0.S || hn.D != h0.D || hn.rows != h0.rows || hn.n != h0.n ||
                hn.k != h0.k || tunedVariant(ops[i + 3 * g.size()]) >= 0)
                break;
            g.push_back(hn);
        }
        auto addr = [](const Tensor &t) { return (intptr_t)t->getRawDataPtr<void *>(); };
        // uniform spacing of weights / biases, outputs back to back (C of group j = C + j * rows * n)
        auto uniform = [&](size_t cnt) {
            const intptr_t dw = addr(g[1].mm->getInputs(1)) - addr(w0);
            const intptr_t db = bias0 ? addr(g[1].mm->getInputs(2)) - addr(bias0) : 0;
            const intptr_t dc = (intptr_t)h0.out->getBytes();
            if (dw % 16 != 0 || db % 2 != 0)
                return false;
            for (size_t j = 1; j < cnt; ++j) {
                if (addr(g[j].mm->getInputs(1)) - addr(w0) != (intptr_t)j * dw)
                    return false;
                if (bias0 && addr(g[j].mm->getInputs(2)) - addr(bias0) != (intptr_t)j * db)
                    return false;
                if (addr(g[j].out) - addr(h0.out) != (intptr_t)j * dc)
                    return false;
            }
            return true;
        };
        size_t cnt = g.size();
        while (cnt >= 2 && !uniform(cnt))
            --cnt;
        // no output may land on anything a later group member still reads
        bool safe = cnt >= 2;
        for (size_t j = 0; safe && j < cnt; ++j)
            for (size_t l = 0; safe && l < cnt; ++l)
                for (const auto &in : g[l].mm->getInputs())
                    if (overlaps(g[j].out, in))
                        safe = false;
        if (safe) {
            const size_t es = a0->getDType().getSize();
            const int64_t strideB = (addr(g[1].mm->getInputs(1)) - addr(w0)) / (intptr_t)es;
            const int64_t strideBias = bias0 ? (addr(g[1].mm->getInputs(2)) - addr(bias0)) / (intptr_t)es : 0;
            if (std::getenv("INFINI_ROCM_FUSION_LOG"))
                fprintf(stderr, "[fusion] headsplit#%zu: %zu projections of one activation grouped into one launch\n", i, cnt);
            ROCM_CALL(infini_rocm_matmul_headsplit(rt, dt, a0->getRawDataPtr<void *>(), w0->getRawDataPtr<void *>(),
                                                   bias0 ? bias0->getRawDataPtr<void *>() : nullptr,
                                                   h0.out->getRawDataPtr<void *>(), (int64_t)cnt, h0.rows, h0.n, h0.k, 0, 0,
                                                   /*strideA*/ 0, strideB, strideBias, 0, bias0 ? 1 : 0, 0, h0.S, h0.D));
            return 3 * cnt;
        }
    }
    OutputRedirect redirect(mm0->getOutput().get(), h0.out->getRawDataPtr<void *>(), (int)h0.S, (int)h0.D);
    launchOne(ops[i]);
    return 3;
}

// MatMul(+bias) -> Gelu (BERT's FFN up-projection), f16 / bf16: the Gelu in the GEMM epilogue (act 5: erf by
// Abramowitz-Stegun 7.1.26, ~18 VALU slots per element instead of erff's ~40). On by default (INFINI_ROCM_FUSE_GELU=0: A/B
// hook). BERT-base bs32 seq512: 5.07 -> 4.73 ms — the separate Gelu pass is a 36.7 us memory-bound kernel per layer, the
// erf evaluations cost the persistent GEMM's store-bound epilogue ~9 us. (Round 1 measured this fusion SLOWER, 5.34 ->
// 5.68 ms, and kept it opt-in: the epilogue selected the activation with a run-time switch that the unrolled tile epilogue
// replicated 128 times, erff and tanhf included; the persistent kernel now has a compile-time Gelu instantiation,
// gemm256p_kernel.h.)
size_t RocmRuntimeObj::tryLaunchMatmulGelu(const OpVec &ops, size_t i) const {
    static const bool enabled = !(std::getenv("INFINI_ROCM_FUSE_GELU") && std::atoi(std::getenv("INFINI_ROCM_FUSE_GELU")) == 0);
    if (!enabled || i + 1 >= ops.size() || ops[i]->getOpType() != OpType::MatMul || ops[i + 1]->getOpType() != OpType::Gelu)
        return 0;
    const Tensor c = ops[i]->getOutput(), out = ops[i + 1]->getOutput();
    const int dt = c->getDTypeIndex();
    if ((dt != INFINI_DT_F16 && dt != INFINI_DT_BF16) || ops[i + 1]->getInputs(0) != c || !soleConsumerIs(c, ops[i + 1]) ||
        c->getBytes() != out->getBytes() || !(c->getDType() == out->getDType()))
        return 0;
    for (const auto &in : ops[i]->getInputs())
        if (overlaps(out, in))
            return 0;
    OutputRedirect redirect(c.get(), out->getRawDataPtr<void *>(), 0, 0, 5);
    launchOne(ops[i]);
    return 2;
}

// Plain MatMuls that multiply the SAME activations by different weights (a decoder block's gate and up projections; q and k
// ahead of their RoPE) are a few operators apart in the list: mm_g, Silu, mm_u, Mul. Each alone leaves the chip part empty
// (Llama-7B at 2048 tokens: 128 or 344 tiles of 256^2 on 256 CUs); as ONE grouped launch (infini_rocm_matmul_grouped: batch
// index = member, zero A stride, the members' weights / outputs at their own uniform distances) they fill it: q + k + v
// 215 -> 169 us, gate + up 330 -> 286 us through the C ABI. A later member runs EARLIER than its place in the list, so:
//   * none of the operators it jumps over may produce (or overwrite) anything it reads;
//   * its output buffer — which the planner handed out for the member's own position — must not overlap anything those
//     operators (or the group) read or write;
//   * a member that another rule wants (head split, Gelu epilogue, copy elision) is left to that rule.
// Members are marked in `launchedAhead` and skipped when the loop reaches them. INFINI_ROCM_GROUP_MATMUL=0 switches it off.
size_t RocmRuntimeObj::tryLaunchGroupedMatmul(const OpVec &ops, size_t i) const {
    static const bool enabled = !(std::getenv("INFINI_ROCM_GROUP_MATMUL") && std::atoi(std::getenv("INFINI_ROCM_GROUP_MATMUL")) == 0);
    if (!enabled || ops[i]->getOpType() != OpType::MatMul)
        return 0;
    auto eligible = [&](size_t j) -> bool {
        if (ops[j]->getOpType() != OpType::MatMul || launchedAhead[j] || tunedVariant(ops[j]) >= 0)
            return false;
        auto mm = as<MatmulObj>(ops[j]);
        const Tensor a = mm->getInputs(0), w = mm->getInputs(1);
        const int dt = a->getDTypeIndex();
        if ((dt != INFINI_DT_F16 && dt != INFINI_DT_BF16) || mm->getTransA() || w->getRank() != 2 || !(w->getDType() == a->getDType()))
            return false;
        if (mm->numInputs() == 3 && !(mm->getInputs(2)->getRank() == 1 && (int)mm->getInputs(2)->size() == w->getDims()[mm->getTransB() ? 0 : 1]))
            return false;
        if (j + 1 < ops.size()) { // someone else's pattern
            const auto t2 = ops[j + 1]->getOpType();
            if ((t2 == OpType::Reshape || t2 == OpType::Flatten || t2 == OpType::Identity || t2 == OpType::Squeeze ||
                 t2 == OpType::Unsqueeze || t2 == OpType::Gelu) && soleConsumerIs(mm->getOutput(), ops[j + 1]))
                return false;
        }
        return true;
    };
    if (!eligible(i))
        return 0;
    auto mm0 = as<MatmulObj>(ops[i]);
    const Tensor a0 = mm0->getInputs(0), w0 = mm0->getInputs(1);
    const Tensor bias0 = mm0->numInputs() == 3 ? mm0->getInputs(2) : nullptr;
    const auto [b0, m0, n0, k0] = mm0->getBMNK();
    std::vector<size_t> members{i};
    std::vector<Tensor> touched; // everything the group and the operators it jumps over read or write
    auto touch = [&](const Operator &o) {
        for (const auto &t : o->getInputs())
            touched.push_back(t);
        for (const auto &t : o->getOutputs())
            touched.push_back(t);
    };
    touch(ops[i]);
    std::vector<Tensor> producedBetween; // outputs of the jumped-over operators
    size_t parkAt = 0;                   // index of the member whose result goes to the workspace (0: none)
    auto wsQuiet = [](const Operator &o) { // kernels that never take the runtime workspace
        const auto t = o->getOpType();
        return t.isUnary() || t == OpType::Silu || t == OpType::Add || t == OpType::Sub || t == OpType::Mul || t == OpType::Div ||
               t == OpType::RoPE || t == OpType::Reshape || t == OpType::Flatten || t == OpType::Identity || t == OpType::Squeeze ||
               t == OpType::Unsqueeze || t == OpType::Transpose;
    };
    constexpr size_t kWindow = 12;
    for (size_t j = i + 1; j < ops.size() && j <= i + kWindow && members.size() < 4; ++j) {
        bool member = false;
        if (eligible(j)) {
            auto mj = as<MatmulObj>(ops[j]);
            const auto [bj, mjm, nj, kj] = mj->getBMNK();
            const Tensor wj = mj->getInputs(1), bj_t = mj->numInputs() == 3 ? mj->getInputs(2) : nullptr;
            member = mj->getInputs(0) == a0 && mj->getTransB() == mm0->getTransB() && wj->getDims() == w0->getDims() && bj == b0 &&
                     mjm == m0 && nj == n0 && kj == k0 && (bj_t != nullptr) == (bias0 != nullptr) &&
                     mj->getOutput()->getDims() == mm0->getOutput()->getDims();
            const Tensor outj = mj->getOutput();
            bool clear = true; // the member's own output buffer is free at the group's position
            for (size_t q = 0; member && clear && q < touched.size(); ++q)
                clear = !overlaps(outj, touched[q]);
            for (size_t q = 0; member && q < producedBetween.size(); ++q)
                for (const auto &in : mj->getInputs())
                    member = member && !overlaps(producedBetween[q], in);
            // The planner usually recycles: mm_u's output sits where mm_g's was (dead once Silu has read it). Then the member's
            // result is PARKED in the workspace and its one consumer — the very next operator, an element-wise / RoPE kernel —
            // reads it from there (parkedFeeds). Two-member groups only (member = batch index needs ONE output stride), nothing
            // between the group and that consumer may use the workspace, and the grouped MatMul itself must not (split-K).
            if (member && !clear) {
                int mayWs = 1;
                const auto [gb, gm, gn, gk] = mj->getBMNK();
                ROCM_CALL(infini_rocm_matmul_may_use_workspace(rt, 2, (int64_t)gb * gm, gn, &mayWs));
                bool quiet = members.size() == 1 && parkAt == 0 && !mayWs && j + 1 < ops.size() && soleConsumerIs(outj, ops[j + 1]) &&
                             wsQuiet(ops[j + 1]) && tunedVariant(ops[j + 1]) < 0;
                for (size_t b = i + 1; quiet && b < j; ++b)
                    quiet = wsQuiet(ops[b]);
                int uses = 0;
                if (quiet)
                    for (const auto &in : ops[j + 1]->getInputs())
                        uses += in == outj;
                if (quiet && uses == 1)
                    parkAt = j;
                else
                    member = false;
            }
        }
        touch(ops[j]);
        if (member) {
            members.push_back(j);
            if (parkAt == j)
                break; // a parked member closes the group
        } else {
            for (const auto &t : ops[j]->getOutputs())
                producedBetween.push_back(t);
        }
    }
    auto addr = [](const Tensor &t) { return (intptr_t)t->getRawDataPtr<void *>(); };
    const intptr_t es = (intptr_t)a0->getDType().getSize();
    const intptr_t cBytes = (intptr_t)mm0->getOutput()->getBytes();
    auto uniform = [&](size_t cnt, intptr_t &dw, intptr_t &db, intptr_t &dc) {
        auto W = [&](size_t q) { return as<MatmulObj>(ops[members[q]])->getInputs(1); };
        auto Bi = [&](size_t q) { return as<MatmulObj>(ops[members[q]])->getInputs(2); };
        auto O = [&](size_t q) { return ops[members[q]]->getOutput(); };
        dw = addr(W(1)) - addr(w0);
        db = bias0 ? addr(Bi(1)) - addr(bias0) : 0;
        dc = addr(O(1)) - addr(O(0));
        if (dw % 16 != 0 || db % es != 0 || dc % 16 != 0 || (dc < cBytes && dc > -cBytes))
            return false;
        for (size_t q = 2; q < cnt; ++q)
            if (addr(W(q)) - addr(w0) != (intptr_t)q * dw || (bias0 && addr(Bi(q)) - addr(bias0) != (intptr_t)q * db) ||
                addr(O(q)) - addr(O(0)) != (intptr_t)q * dc)
                return false;
        return true;
    };
    size_t cnt = members.size();
    intptr_t dw = 0, db = 0, dc = 0;
    void *park = nullptr;
    if (parkAt) { // exactly two members: the second one's result goes to the workspace
        park = getWorkspace((size_t)cBytes);
        auto W1 = as<MatmulObj>(ops[members[1]])->getInputs(1);
        dw = addr(W1) - addr(w0);
        db = bias0 ? addr(as<MatmulObj>(ops[members[1]])->getInputs(2)) - addr(bias0) : 0;
        dc = (intptr_t)park - addr(mm0->getOutput());
        if (cnt != 2 || dw % 16 != 0 || db % es != 0 || dc % 16 != 0 || (dc < cBytes && dc > -cBytes))
            return 0;
    } else {
        while (cnt >= 2 && !uniform(cnt, dw, db, dc))
            --cnt;
        if (cnt < 2)
            return 0;
    }
    // rows: the batch folds into m when the weight is shared (rank-2 w) and A is dense
    const int64_t rows = (int64_t)b0 * m0;
    if (std::getenv("INFINI_ROCM_FUSION_LOG"))
        fprintf(stderr, "[fusion] matmul#%zu: %zu MatMuls of one activation grouped into one launch (members up to #%zu)\n", i, cnt,
                members[cnt - 1]);
    ROCM_CALL(infini_rocm_matmul_grouped(rt, a0->getDTypeIndex(), a0->getRawDataPtr<void *>(), w0->getRawDataPtr<void *>(),
                                         bias0 ? bias0->getRawDataPtr<void *>() : nullptr,
                                         mm0->getOutput()->getRawDataPtr<void *>(), (int64_t)cnt, rows, n0, k0, 0,
                                         mm0->getTransB() ? 1 : 0, /*strideA*/ 0, dw / es, dc / es, db / es, 0, bias0 ? 1 : 0, 0, 0, 0));
    for (size_t q = 1; q < cnt; ++q)
        launchedAhead[members[q]] = 1;
    if (parkAt) {
        parkedFeeds[parkAt + 1] = ParkedFeed{ops[parkAt]->getOutput().get(), park};
        ++parkedCount;
    }
    return 1;
}

// RoPE -> Reshape([B, S, H, D]) -> Transpose(0, 2, 1, 3): the rotary embedding of a decoder's q / k followed by their head split
// (rope.cu, reshape.cc, transpose.cc: three launches, two extra passes) as one pass with a head-split store
// (infini_rocm_rope_headsplit). Head dim 128 / theta 1e4 as the reference hard-codes them (rope.cc:25). The input may be a
// grouped MatMul's result parked in the workspace (parkedFeeds): this rule resolves it itself, launchAll tries it first.
size_t RocmRuntimeObj::tryLaunchRopeHeadSplit(const OpVec &ops, size_t i) const {
    static const bool enabled = !(std::getenv("INFINI_ROCM_FUSE_ROPE_SPLIT") && std::atoi(std::getenv("INFINI_ROCM_FUSE_ROPE_SPLIT")) == 0);
    if (!enabled || i + 2 >= ops.size() || ops[i]->getOpType() != OpType::RoPE || ops[i + 1]->getOpType() != OpType::Reshape ||
        ops[i + 2]->getOpType() != OpType::Transpose || launchedAhead[i + 1] || launchedAhead[i + 2])
        return 0;
    const Tensor pos = ops[i]->getInputs(0), x = ops[i]->getInputs(1), y = ops[i]->getOutput();
    const Tensor r = ops[i + 1]->getOutput(), out = ops[i + 2]->getOutput();
    auto tr = as<TransposeObj>(ops[i + 2]);
    if (ops[i + 1]->getInputs(0) != y || tr->getInputs(0) != r || !soleConsumerIs(y, ops[i + 1]) || !soleConsumerIs(r, ops[i + 2]))
        return 0;
    const auto &xd = x->getDims(), &rd = r->getDims();
    const auto perm = tr->getPermute();
    if (xd.size() != 3 || rd.size() != 4 || perm.size() != 4 || perm[0] != 0 || perm[1] != 2 || perm[2] != 1 || perm[3] != 3 ||
        rd[0] != xd[0] || rd[1] != xd[1] || rd[3] != 128 || (long)rd[2] * rd[3] != xd[2] || pos->getDims().size() != 2 ||
        pos->getDims()[1] != xd[1] || !(out->getDType() == x->getDType()) || out->getBytes() != x->getBytes())
        return 0;
    const void *xp = x->getRawDataPtr<void *>();
    auto pf = parkedFeeds.find(i);
    if (pf != parkedFeeds.end()) {
        if (pf->second.tensor != x.get())
            return 0;
        xp = pf->second.ptr; // the tensor's own buffer holds something else right now: only the parked copy counts
    } else if (overlaps(out, x)) {
        return 0;
    }
    if (overlaps(out, pos))
        return 0;
    ROCM_CALL(infini_rocm_rope_headsplit(rt, x->getDTypeIndex(), pos->getDTypeIndex(), pos->getRawDataPtr<void *>(), xp,
                                         out->getRawDataPtr<void *>(), (int64_t)xd[0] * xd[1], xd[2], 128, 10000.0f, xd[1]));
    if (pf != parkedFeeds.end())
        parkedFeeds.erase(pf);
    return 3;
}

// producer -> Reshape | Flatten | Identity | Squeeze | Unsqueeze: the reference runs these as a device memcpy
// (CopyCuda, reshape.cc:4-21); here the producer writes straight into the copy's output buffer (BERT: the three
// projections before the head split and the head merge, 48 copies of 25 MB per forward). The producer runs through its
// normal kernel (and perf record) with its output tensor redirected, so nothing about its numerics changes.
size_t RocmRuntimeObj::tryLaunchIntoReshape(const OpVec &ops, size_t i) const {
    static const bool enabled = !(std::getenv("INFINI_ROCM_FUSE_RESHAPE") && std::atoi(std::getenv("INFINI_ROCM_FUSE_RESHAPE")) == 0);
    if (!enabled || i + 1 >= ops.size()) // INFINI_ROCM_FUSE_RESHAPE=0: A/B hook
        return 0;
    const Operator &op = ops[i], &next = ops[i + 1];
    const auto nt = next->getOpType(), type = op->getOpType();
    if (!(nt == OpType::Reshape || nt == OpType::Flatten || nt == OpType::Identity || nt == OpType::Squeeze ||
          nt == OpType::Unsqueeze))
        return 0;
    // producers whose kernels only WRITE their output (no in-place state, no multi-output, no collectives)
    if (!(type == OpType::MatMul || type == OpType::Transpose || type == OpType::Add || type == OpType::Sub ||
          type == OpType::Mul || type == OpType::Div || type == OpType::Relu || type == OpType::Gelu ||
          type == OpType::Silu || type == OpType::Sigmoid || type == OpType::Tanh || type == OpType::Softmax ||
          type == OpType::LayerNormalization || type == OpType::RMSNorm || type == OpType::Gather || type == OpType::RoPE))
        return 0;
    if (op->numOutputs() != 1 || next->numOutputs() != 1)
        return 0;
    const Tensor mid = op->getOutput(), out = next->getOutput();
    if (next->getInputs(0) != mid || !soleConsumerIs(mid, next) || mid->getBytes() != out->getBytes() ||
        !(mid->getDType() == out->getDType()))
        return 0;
    for (const auto &in : op->getInputs())
        if (overlaps(out, in)) // the producer would overwrite what it is still reading
            return 0;
    OutputRedirect redirect(mid.get(), out->getRawDataPtr<void *>());
    launchOne(op);
    return 2;
}

size_t RocmRuntimeObj::tryLaunchFusedRules(const OpVec &ops, size_t i) const {
    const Operator &op = ops[i];
    const auto type = op->getOpType();
    if (type == OpType::MatMul)
        return tryLaunchFusedAttention(ops, i);
    if (type == OpType::Conv) {
        auto conv = as<ConvObj>(op);
        const Tensor x = conv->getInputs(0), w = conv->getInputs(1);
        const int f = conv->getOutput()->getDims()[1];
        // candidate chains, each one op longer than the previous: conv [+ bias] [+ residual] [+ relu]
        struct Cand {
            size_t used;
            Tensor last, bias, res;
            int act;
        };
        std::vector<Cand> cands;
        Cand cur{1, conv->getOutput(), nullptr, nullptr, 0};
        auto nextIs = [&](OpType t) {
            return i + cur.used < ops.size() && ops[i + cur.used]->getOpType() == t && soleConsumerIs(cur.last, ops[i + cur.used]);
        };
        auto otherOf = [&](const Operator &o) { return o->getInputs(0) == cur.last ? o->getInputs(1) : o->getInputs(0); };
        if (nextIs(OpType::Add)) {
            const Operator &add = ops[i + cur.used];
            const Tensor other = otherOf(add);
            if (other != cur.last && isChannelBias(other, f) && other->getDType() == cur.last->getDType() &&
                add->getOutput()->getDims() == cur.last->getDims()) {
                cur.bias = other;
                cur.last = add->getOutput();
                ++cur.used;
                cands.push_back(cur);
            }
        }
        // The residual join rides in the conv epilogue where the LDS-staged epilogue serves it (conv_s1.hip: the residual
        // is fetched in the same 128-byte row segments as the stores; even output planes, f16 / bf16). With the earlier
        // direct epilogue (32-byte segments per filter row) the fused form was slower than conv + one ADD_RELU pass, which
        // is still what odd planes (7x7) and fp32 get. INFINI_ROCM_FUSE_RES=0 / =1 forces it off / on for every shape.
        static const int fuseResEnv = std::getenv("INFINI_ROCM_FUSE_RES") ? std::atoi(std::getenv("INFINI_ROCM_FUSE_RES")) : -1;
        const auto &od = conv->getOutput()->getDims();
        const bool fuseRes = fuseResEnv >= 0 ? fuseResEnv == 1
                                             : (od.size() == 4 && ((long)od[2] * od[3]) % 2 == 0 && conv->getNumGroups() == 1 &&
                                                !(x->getDType() == DataType::Float32) && !(x->getDType() == DataType::Double));
        if (fuseRes && cur.bias && nextIs(OpType::Add)) { // residual join: the tail of a ResNet bottleneck
            const Operator &add2 = ops[i + cur.used];
            const Tensor other = otherOf(add2);
            if (other != cur.last && other->getDims() == cur.last->getDims() && other->getDType() == cur.last->getDType()) {
                cur.res = other;
                cur.last = add2->getOutput();
                ++cur.used;
                cands.push_back(cur);
            }
        }
        if (nextIs(OpType::Relu)) {
            cur.act = 1;
            cur.last = ops[i + cur.used]->getOutput();
            ++cur.used;
            cands.push_back(cur);
        }
        // longest chain whose output buffer is safe to write while the conv still reads its inputs
        static const bool fusionLog = std::getenv("INFINI_ROCM_FUSION_LOG") != nullptr; // diagnostics: which chain each conv got
        for (auto it = cands.rbegin(); it != cands.rend(); ++it) {
            const Cand &c = *it;
            if (fusionLog)
                fprintf(stderr, "[fusion] conv#%zu [%d,%d,%d,%d]: chain %zu (bias %d res %d act %d) out-on-x %d out-on-w %d out-on-res %d\n", i,
                        (int)od[0], (int)od[1], (int)od[2], (int)od[3], c.used, c.bias != nullptr, c.res != nullptr, c.act,
                        (int)overlaps(c.last, x), (int)overlaps(c.last, w), (int)(c.res && overlaps(c.last, c.res)));
            // the residual is read at exactly the position that is written: it may be the output buffer itself
            const bool resHazard = c.res && overlaps(c.last, c.res) &&
                                   !(c.res->getRawDataPtr<void *>() == c.last->getRawDataPtr<void *>() &&
                                     c.res->getDims() == c.last->getDims());
            const auto [n, ch, h, wd, ff, r, s] = conv->getNCHWFRS();
            const auto [ph, pw, sh, sw, dh, dw] = conv->getPadStrideDilation();
            // The planner likes to put the chain's output on the conv's own input (dead after the conv in the unfused
            // graph). When that input is small next to the output — the 64 -> 256 expansions of ResNet's first stage: 51 MB
            // in, 205 MB out — the conv reads a copy of it from the workspace instead of giving up the tail: one 2 x 51 MB
            // copy instead of a lone 2 x 205 MB ReLU / bias pass. Unit-stride only (a strided conv keeps its phase planes
            // at the workspace base, a long-K pointwise layer may run as a split-K GEMM with partial planes there); the
            // copy sits behind the conv's own packed-weight area.
            const bool onX = overlaps(c.last, x);
            static const bool bridgeOn = !(std::getenv("INFINI_ROCM_BRIDGE_X") && std::atoi(std::getenv("INFINI_ROCM_BRIDGE_X")) == 0);
            const bool bridgeX = bridgeOn && onX && sh == 1 && sw == 1 && conv->getNumGroups() == 1 && ch < 1024 &&
                                 (size_t)x->getBytes() * 3 <= (size_t)c.last->getBytes();
            if (!(c.last->getDType() == x->getDType()) || (onX && !bridgeX) || overlaps(c.last, w) ||
                (c.bias && overlaps(c.last, c.bias)) || resHazard)
                continue;
            const void *xptr = x->getRawDataPtr<void *>();
            if (bridgeX) {
                const size_t wArea = (((size_t)ff * ch * r * s * 2 + 4096) + 255) & ~(size_t)255;
                char *ws = (char *)getWorkspace(wArea + x->getBytes());
                ROCM_CALL(infini_rocm_copy_inside(rt, ws + wArea, xptr, x->getBytes()));
                xptr = ws + wArea;
                ++bridgedCount;
                if (fusionLog)
                    fprintf(stderr, "[fusion] conv#%zu: input bridged through the workspace (%zu bytes)\n", i, (size_t)x->getBytes());
            }
            // a tuned Conv (h.tune(): ConvRocm::tune) keeps its kernel choice when the tail is folded into it
            struct VariantScope {
                infiniRocmRuntime_t rt;
                bool set;
                VariantScope(infiniRocmRuntime_t rt, int v) : rt(rt), set(v >= 0) {
                    if (set)
                        ROCM_CALL(infini_rocm_conv2d_set_variant(rt, v));
                }
                ~VariantScope() {
                    if (set)
                        (void)infini_rocm_conv2d_set_variant(rt, -1);
                }
            } scope(rt, tunedVariant(op));
            ConstWeightsScope constWeights(rt, w); // graph weights: pack once, cache (rocm_runtime.h)
            ROCM_CALL(infini_rocm_conv2d_res(rt, x->getDTypeIndex(), xptr, w->getRawDataPtr<void *>(),
                                             c.bias ? c.bias->getRawDataPtr<void *>() : nullptr,
                                             c.res ? c.res->getRawDataPtr<void *>() : nullptr, c.last->getRawDataPtr<void *>(),
                                             n, ch, h, wd, ff, r, s, ph, pw, sh, sw, dh, dw, conv->getNumGroups(), c.act));
            return c.used;
        }
        return 0;
    }
    // Silu(a) -> Mul(., b): the gate of a gated MLP as one pass (infini_rocm_silu_mul; bit-identical: the Silu value is rounded
    // as its own kernel would have stored it). Operators between the two that a grouped launch already ran (the up projection,
    // rocm_fusion.cc::tryLaunchGroupedMatmul) are skipped over; b may be a result parked in the workspace by that launch.
    if (type == OpType::Silu) {
        static const bool swigluOn = !(std::getenv("INFINI_ROCM_FUSE_SWIGLU") && std::atoi(std::getenv("INFINI_ROCM_FUSE_SWIGLU")) == 0);
        size_t m = i + 1;
        while (m < ops.size() && launchedAhead[m])
            ++m;
        if (swigluOn && m < ops.size() && ops[m]->getOpType() == OpType::Mul && soleConsumerIs(op->getOutput(), ops[m])) {
            const Tensor a = op->getInputs(0), sOut = op->getOutput(), out = ops[m]->getOutput();
            const Tensor m0 = ops[m]->getInputs(0), m1 = ops[m]->getInputs(1);
            const Tensor b = m0 == sOut ? m1 : m0;
            const int dt = a->getDTypeIndex();
            auto safe = [&](const Tensor &u) { // element-wise: exactly in place is fine, a partial overlap is not
                return !overlaps(out, u) || (u->getRawDataPtr<void *>() == out->getRawDataPtr<void *>() && u->getDims() == out->getDims());
            };
            const void *bp = b->getRawDataPtr<void *>();
            auto pf = parkedFeeds.find(m);
            const bool bParked = pf != parkedFeeds.end() && pf->second.tensor == b.get();
            if (bParked)
                bp = pf->second.ptr;
            if (b != sOut && (m0 == sOut) != (m1 == sOut) && (dt == INFINI_DT_F16 || dt == INFINI_DT_BF16 || dt == INFINI_DT_F32) &&
                a->getDims() == out->getDims() && b->getDims() == out->getDims() && b->getDType() == a->getDType() &&
                out->getDType() == a->getDType() && safe(a) && (bParked || safe(b)) &&
                (((uintptr_t)a->getRawDataPtr<void *>() | (uintptr_t)bp | (uintptr_t)out->getRawDataPtr<void *>()) & 15) == 0 &&
                (pf == parkedFeeds.end() || bParked)) {
                ROCM_CALL(infini_rocm_silu_mul(rt, dt, a->getRawDataPtr<void *>(), bp, out->getRawDataPtr<void *>(), (int64_t)out->size()));
                if (bParked)
                    parkedFeeds.erase(pf);
                return m - i + 1; // everything in between already ran
            }
        }
    }
    // Relu -> MaxPool: max and relu commute (bit-identical); the stem of every ResNet
    if (type == OpType::Relu && i + 1 < ops.size() && ops[i + 1]->getOpType() == OpType::MaxPool &&
        soleConsumerIs(op->getOutput(), ops[i + 1])) {
        auto pool = as<PoolingObj>(ops[i + 1]);
        const Tensor x = op->getInputs(0), out = pool->getOutput();
        if (!overlaps(out, x)) {
            const auto [n, c, h, w, kh, kw] = pool->getNCHWRS();
            const auto [ph, pw, sh, sw, dh, dw] = pool->getPadStrideDilation();
            ROCM_CALL(infini_rocm_pool2d_relu(rt, 0, x->getDTypeIndex(), x->getRawDataPtr<void *>(), out->getRawDataPtr<void *>(),
                                              n, c, h, w, kh, kw, dh, dw, ph, pw, sh, sw, pool->getCeilMode(), 1));
            return 2;
        }
    }
    // Add(a, b) (same extents) -> LayerNormalization over the last axis / RMSNorm: one pass
    if (type == OpType::Add && i + 1 < ops.size() && soleConsumerIs(op->getOutput(), ops[i + 1]) &&
        (ops[i + 1]->getOpType() == OpType::LayerNormalization || ops[i + 1]->getOpType() == OpType::RMSNorm)) {
        const Operator &nrm = ops[i + 1];
        const Tensor a = op->getInputs(0), b = op->getInputs(1), t = op->getOutput(), out = nrm->getOutput();
        const auto &td = t->getDims();
        const bool rms = nrm->getOpType() == OpType::RMSNorm;
        bool ok = a->getDims() == td && b->getDims() == td && a->getDType() == b->getDType() && nrm->getInputs(0) == t;
        float eps = 1e-5f; // RMSNorm: hard-coded in the reference (rms_norm.cu:46)
        Tensor scale = nrm->getInputs(1), bias = nullptr;
        if (ok && !rms) {
            auto ln = as<LayerNormObj>(nrm);
            ok = ln->getAxis() == (int)td.size() - 1;
            eps = ln->getEps();
            if (ln->numInputs() == 3)
                bias = ln->getInputs(2);
        }
        if (ok) {
            auto hazard = [&](const Tensor &u) { // row-wise in place over an operand of identical extent is safe
                const bool inPlace = u->getRawDataPtr<void *>() == out->getRawDataPtr<void *>() && u->getDims() == td;
                return overlaps(out, u) && !inPlace;
            };
            if (!hazard(a) && !hazard(b) && !overlaps(out, scale) && !(bias && overlaps(out, bias))) {
                const int64_t n = td.back(), outer = (int64_t)t->size() / n;
                ROCM_CALL(infini_rocm_add_norm(rt, t->getDTypeIndex(), rms ? 1 : 0, a->getRawDataPtr<void *>(),
                                               b->getRawDataPtr<void *>(), scale->getRawDataPtr<void *>(),
                                               bias ? bias->getRawDataPtr<void *>() : nullptr, out->getRawDataPtr<void *>(),
                                               outer, n, (int64_t)scale->size(), bias ? (int64_t)bias->size() : 0, eps));
                return 2;
            }
        }
    }
    // Add(x, per-channel bias) -> Add(., identity) [-> Relu]: the bottleneck tail when the bias could not ride in the conv
    if (type == OpType::Add && i + 1 < ops.size() && ops[i + 1]->getOpType() == OpType::Add &&
        soleConsumerIs(op->getOutput(), ops[i + 1])) {
        const Tensor t = op->getOutput();
        const Tensor a0 = op->getInputs(0), a1 = op->getInputs(1);
        const auto &td = t->getDims();
        Tensor xin = nullptr, bias = nullptr;
        if (td.size() >= 2 && isChannelBiasOf(a1, td) && a0->getDims() == td) { xin = a0; bias = a1; }
        else if (td.size() >= 2 && isChannelBiasOf(a0, td) && a1->getDims() == td) { xin = a1; bias = a0; }
        if (xin) {
            const Operator &add2 = ops[i + 1];
            const Tensor res = add2->getInputs(0) == t ? add2->getInputs(1) : add2->getInputs(0);
            Tensor out = add2->getOutput();
            size_t used = 2;
            int relu = 0;
            if (res != t && res->getDims() == td && res->getDType() == t->getDType() && bias->getDType() == t->getDType()) {
                if (i + 2 < ops.size() && ops[i + 2]->getOpType() == OpType::Relu && soleConsumerIs(out, ops[i + 2])) {
                    out = ops[i + 2]->getOutput();
                    relu = 1;
                    used = 3;
                }
                auto hazard = [&](const Tensor &u) {
                    const bool inPlace = u->getRawDataPtr<void *>() == out->getRawDataPtr<void *>() && u->getDims() == td;
                    return overlaps(out, u) && !inPlace;
                };
                if (!hazard(xin) && !hazard(res) && !overlaps(out, bias)) {
                    int64_t inner = 1;
                    for (size_t d = 2; d < td.size(); ++d)
                        inner *= td[d];
                    ROCM_CALL(infini_rocm_bias_residual(rt, t->getDTypeIndex(), xin->getRawDataPtr<void *>(),
                                                        bias->getRawDataPtr<void *>(), res->getRawDataPtr<void *>(),
                                                        out->getRawDataPtr<void *>(), td[0], td[1], inner, relu));
                    return used;
                }
            }
        }
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
