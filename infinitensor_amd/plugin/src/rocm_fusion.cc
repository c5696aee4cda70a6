// Launch planning for Device::ROCM. The reference executes one kernel per operator (cuda_runtime.cc:136-170); on
// MI355X the element-wise tails of a convolution / GEMM are pure HBM traffic, so before a graph is launched the runtime
// PLANS it: operators that can run as one kernel become one plan item, everything else stays one item per operator.
// The plan is then executed item by item (rocm_runtime.cc::executePlan); nothing of the graph is modified.
//
// The chains are matched on the graph the reference's ONNX front-end actually builds (pyinfinitensor/onnx.py), i.e.
//   * a Conv's bias arrives as   Conv -> Reshape(bias, [1, F, 1, 1]) -> Add           (onnx.py:159-190)
//   * a linear layer as          MatMul(x, W) [no bias, no transpose] -> Add(bias)    (onnx.py:280-290; only Gemm,
//     :291-311, carries bias / transposes)
//   * Q.K^T as                   Transpose(K) -> MatMul(Q, K^T)
//   * opset < 17 LayerNorm as    ReduceMean, Sub, Pow, ReduceMean, Add, Sqrt, Div, Mul, Add    (nine operators)
//   * opset < 20 Gelu as         Div, Erf, Add, Mul, Mul
// and in whatever operator order the exporter chose: a matcher follows the CONSUMER of the chain's tensor, not the next
// operator in the list (the front-end's Reshape of a conv bias sits between the Conv and its Add; a transformer's q
// projection is issued before k and v but reshaped after them).
//
// Rules (every intermediate tensor has exactly one consumer and is not a graph output; same dtype throughout):
//   Conv -> Add(per-channel bias) [-> Add(identity)] [-> Relu]     =>  conv2d_res(bias, residual, act)
//   MatMul [-> Add(row bias)] [-> Gelu] [-> Reshape [B,S,H,D] -> Transpose(0,2,1,3)  |  -> Reshape-family copy]
//                                                                  =>  one GEMM: bias / Gelu in the epilogue, head-split
//        or redirected store; two or three such projections of ONE activation (q, k, v) => one grouped launch
//   [Transpose(K)] -> MatMul(Q, K^T) [-> Div|Mul(scalar)] [-> Add(mask)] -> Softmax(last axis) -> MatMul(P, V)
//        [-> Transpose(0,2,1,3) -> Reshape]                        =>  attention (csrc/attention.hip): the score matrix is
//        never written. K^T may be MatMul's transB, a Transpose(0,1,3,2) of the head-split K (absorbed: the kernel
//        reads K), or a Transpose(0,2,3,1) of K's [B,S,H,D] view (the producer stores K head-major instead)
//   ReduceMean, Sub, Pow(2), ReduceMean, Add(eps), Sqrt, Div, Mul(gamma), Add(beta)   =>  layer_norm  (one pass)
//   Div(x, sqrt 2), Erf, Add(1), Mul(x, .), Mul(., 0.5)  (any operand order)          =>  Gelu (unary kernel or epilogue)
//   Add -> LayerNormalization(last axis) | RMSNorm | decomposed LayerNorm             =>  add_norm
//   Add(bias) -> Add(identity) [-> Relu]  =>  bias_residual;     Add -> Relu  =>  binary(ADD_RELU)
//   Relu -> MaxPool  =>  pool2d_relu;     Silu(a) -> Mul(., b)  =>  silu_mul;     RoPE -> Reshape -> Transpose  =>  rope_headsplit
//   producer -> Reshape-family copy       =>  the producer writes into the copy's output
//   plain MatMuls of one activation a few operators apart (gate / up) => one grouped launch (run AHEAD of their place)
//   Reshape-family operators on a weight (the conv bias Reshape)  =>  not launched: a reshape does not change the bytes,
//        the consumer reads the weight itself
//
// When a fused item runs. An item made of operators at positions p0 < p1 < ... < pk runs at pk (it "sinks"): its final
// tensor is written exactly when the unfused graph would write it, so the memory planner's reuse of dead storage stays
// valid for everything else. What moves is the time the item READS the inputs of its earlier members; the planner proves
// for each such tensor that no operator between its original reader and pk writes memory overlapping it (`survives`) —
// else the chain is cut before that member. Results of earlier members that ARE materialised later than planned (k and v
// of a grouped q / k / v launch) must have no reader in between (`unreadUntil`). The one rule that runs operators AHEAD of
// their place (grouped gate / up) checks the opposite: the early write must not land on anything the jumped-over
// operators — or a sunk item's late reads — still use.
// In-kernel hazards: the final buffer must not overlap an input of the fused kernel (exception: element-wise kernels
// exactly in place over a same-extent input).
// Numerics: fp32 chains run the same operations in the same order (bit-identical); f16 / bf16 round once instead of after
// every operator; the decomposed LayerNorm / Gelu forms compute in fp32 what the nine / five operators round per step.
// INFINI_ROCM_FUSION=0 or RocmRuntimeObj::setFusion(false) restores one kernel per operator.
//
// File layout (round 4). This file holds the planner's CORE: the plan driver (run / tryRules / planTwice), tensor facts
// (onlyUser, aliasRoot, persistent, scalarOf) and the HAZARD MODULE every rule must go through — overlaps, survives,
// unreadUntil, readsSurvive, noteLateReads / lateReads, the ForwardMap, emit (claimed / writeAt bookkeeping). The rules
// themselves are class members kept in one file per family, included inside the class body below:
//   rocm_fusion_rules_decomposed.inc   primitive-operator LayerNorm / Gelu (matchers + rules)
//   rocm_fusion_rules_attention.inc    the fused attention launch and the Transpose(K) forms
//   rocm_fusion_rules_matmul.inc       MatMul chains, buffer forwarding, grouped launches, row-parallel all-reduce overlap
//   rocm_fusion_rules_conv.inc         Conv chains, the fused stem, input bridging, forwarding
//   rocm_fusion_rules_elementwise.inc  element-wise / row-wise pairs, RoPE head split, producers into copies
#include "core/perf_engine.h"
#include "operators/conv.h"
#include "operators/element_wise.h"
#include "operators/layer_norm.h"
#include "operators/matmul.h"
#include "operators/pooling.h"
#include "operators/reduce.h"
#include "operators/rms_norm.h"
#include "operators/softmax.h"
#include "operators/transpose.h"
#include "operators/unary.h"
#include "rocm/rocm_perf.h"
#include "rocm/rocm_runtime.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <optional>
#include <string>
#include <unordered_map>

namespace infini {

thread_local RocmRuntimeObj::LaunchOverrides RocmRuntimeObj::overrides;
thread_local const RocmRuntimeObj::ForwardMap *RocmRuntimeObj::forwards = nullptr;

namespace {
using PlanItem = RocmRuntimeObj::PlanItem;
using LaunchPlan = RocmRuntimeObj::LaunchPlan;

struct OverrideScope { // RAII: overrides never outlive one launch
    OverrideScope() { RocmRuntimeObj::overrides = RocmRuntimeObj::LaunchOverrides(); }
    ~OverrideScope() { RocmRuntimeObj::overrides = RocmRuntimeObj::LaunchOverrides(); }
    void redirect(const TensorObj *t, void *p) {
        auto &o = RocmRuntimeObj::overrides;
        const int s = o.tensor[0] ? 1 : 0;
        IT_ASSERT(o.tensor[s] == nullptr, "more than two tensor overrides in one launch");
        o.tensor[s] = t;
        o.ptr[s] = p;
    }
};

bool envOn(const char *name) { // default on; NAME=0 is the A/B hook
    const char *e = std::getenv(name);
    return !(e && std::atoi(e) == 0);
}
// where a tensor's value lives: its own buffer, unless the plan forwarded it (RocmRuntimeObj::ForwardMap) — valid while a
// plan is being made (the planner installs its map) and while it runs (executePlan does)
void *dataPtr(const Tensor &t) {
    if (RocmRuntimeObj::forwards) {
        auto it = RocmRuntimeObj::forwards->find(t.get());
        if (it != RocmRuntimeObj::forwards->end())
            return it->second;
    }
    return t->getRawDataPtr<void *>();
}
uintptr_t addrOf(const Tensor &t) { return reinterpret_cast<uintptr_t>(dataPtr(t)); }
bool overlaps(const Tensor &a, const Tensor &b) {
    const auto pa = addrOf(a), pb = addrOf(b);
    return pa < pb + b->getBytes() && pb < pa + a->getBytes();
}
bool samePlace(const Tensor &a, const Tensor &b) { return addrOf(a) == addrOf(b) && a->getDims() == b->getDims(); }
bool isCopyLike(OpType t) {
    return t == OpType::Reshape || t == OpType::Flatten || t == OpType::Identity || t == OpType::Squeeze || t == OpType::Unsqueeze;
}
bool isChannelBias(const Shape &d, int f) {
    if (d.size() == 4)
        return d[0] == 1 && d[1] == f && d[2] == 1 && d[3] == 1;
    if (d.size() == 3)
        return d[0] == f && d[1] == 1 && d[2] == 1;
    return false;
}
// `bd` is a per-channel bias of a tensor with dims `d` ([N, C, ...]): [1, C, 1, ...] or [C, 1, ...]
bool isChannelBiasOf(const Shape &bd, const Shape &d) {
    int64_t sz = 1;
    for (int v : bd)
        sz *= v;
    if (d.size() < 2 || sz != d[1])
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
// `bd` broadcasts to one row of n elements: [n], [1, n], [1, 1, n], ...
bool isRowVector(const Shape &bd, int n) {
    if (bd.empty() || bd.back() != n)
        return false;
    for (size_t i = 0; i + 1 < bd.size(); ++i)
        if (bd[i] != 1)
            return false;
    return true;
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
bool isHalf(int dt) { return dt == INFINI_DT_F16 || dt == INFINI_DT_BF16; }
bool permIs(const std::vector<int> &p, int a, int b, int c, int d) {
    return p.size() == 4 && p[0] == a && p[1] == b && p[2] == c && p[3] == d;
}
double halfToDouble(uint16_t h) {
    const int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    double v = e == 0 ? std::ldexp((double)m, -24) : (e == 31 ? (m ? NAN : INFINITY) : std::ldexp((double)(m | 1024), e - 25));
    return s ? -v : v;
}
} // namespace

class FusionPlanner {
  public:
    // silent: operators a previous planning pass found to write nothing of their own (members of fused items whose outputs
    // are not materialised). `survives` then does not count their planned output buffers as written — which is what lets a
    // forwarded buffer outlive the operators BEHIND the chain that will themselves be folded away. A speculation: the caller
    // verifies it against this pass's own result (silentHolds) and falls back to the first pass's plan otherwise.
    FusionPlanner(const RocmRuntimeObj *R, const OpVec &ops, bool fusion, const std::vector<char> *silent = nullptr)
        : R(R), rt(R ? R->handle() : nullptr), ops(ops), n(ops.size()), fusion(fusion), silent(silent), claimed(ops.size(), 0),
          writeAt(ops.size(), -1), log(std::getenv("INFINI_ROCM_FUSION_LOG") != nullptr),
          fwd(std::make_shared<RocmRuntimeObj::ForwardMap>()) {
        posOf.reserve(n * 2);
        for (size_t i = 0; i < n; ++i)
            posOf[ops[i].get()] = i;
        RocmRuntimeObj::forwards = fwd.get(); // addrOf / dataPtr see the forwarded buffers while the plan is made
    }
    ~FusionPlanner() { RocmRuntimeObj::forwards = nullptr; }

    LaunchPlan run() {
        for (size_t i = 0; i < n; ++i) {
            if (claimed[i])
                continue;
            bool done = false;
            if (auto pk = parked.find(i); pk != parked.end()) {
                done = fusion && planParkedConsumer(i, pk->second);
                if (!done) { // plain launch, the parked operand read from the workspace
                    const ParkedFeed pf = pk->second;
                    const Operator op = ops[i];
                    const RocmRuntimeObj *r = R;
                    emit(i, {i}, "", false, [r, op, pf] {
                        OverrideScope s;
                        s.redirect(pf.tensor, r->getWorkspace(pf.bytes));
                        r->launchOne(op);
                    });
                    done = true;
                }
            }
            if (!done && fusion)
                done = tryRules(i);
            if (!done) {
                const Operator op = ops[i];
                const RocmRuntimeObj *r = R;
                emit(i, {i}, "", false, [r, op] { r->launchOne(op); });
            }
        }
        std::stable_sort(items.begin(), items.end(), [](const PlanItem &a, const PlanItem &b) { return a.slot < b.slot; });
        LaunchPlan plan;
        plan.items = std::move(items);
        plan.forwarded = fwd;
        return plan;
    }

    // operators that wrote nothing of their own in this pass
    std::vector<char> silentOps() const {
        std::vector<char> v(n, 0);
        for (size_t p = 0; p < n; ++p)
            v[p] = claimed[p] && writeAt[p] < 0;
        return v;
    }
    bool silentHolds() const {
        if (!silent)
            return true;
        for (size_t p = 0; p < n; ++p)
            if ((*silent)[p] && !(claimed[p] && writeAt[p] < 0)) {
                if (log)
                    fprintf(stderr, "[fusion] two-pass: operator %zu was assumed silent but %s\n", p, claimed[p] ? "writes" : "is not planned");
                return false;
            }
        return true;
    }
    size_t forwardedTensors() const { return fwd->size(); }

  private:
    const RocmRuntimeObj *R; // nullptr: dry run (describeFusionPlan on a non-ROCM runtime)
    infiniRocmRuntime_t rt;
    const OpVec &ops;
    const size_t n;
    const bool fusion;
    const std::vector<char> *silent;
    std::vector<char> claimed;
    // for a claimed operator: the slot at which ITS output tensor is really written (-1: never — an intermediate that is
    // not materialised, a grouped result parked in the workspace)
    std::vector<long> writeAt;
    std::unordered_map<const OperatorObj *, size_t> posOf;
    std::vector<PlanItem> items;
    const bool log;
    std::shared_ptr<RocmRuntimeObj::ForwardMap> fwd;
    struct LateRead { // a sunk item reads `t` at `to` instead of `from`
        size_t from, to;
        Tensor t;
    };
    std::vector<LateRead> lateReads;
    struct ParkedFeed { // the operator at the key position reads `tensor` from the workspace (getWorkspace(bytes))
        const TensorObj *tensor;
        size_t bytes;
    };
    std::map<size_t, ParkedFeed> parked;

    // ---- graph helpers -----------------------------------------------------------------------------------------
    size_t pos(const Operator &op) const { return posOf.at(op.get()); }
    // the one operator that reads `t` (nullptr: several readers, none, or a graph output that must exist in memory)
    Operator onlyUser(const Tensor &t) const {
        if (t->isOutput())
            return nullptr;
        const auto tg = t->getTargets();
        if (tg.size() != 1)
            return nullptr;
        auto it = posOf.find(tg[0].get());
        if (it == posOf.end() || claimed[it->second])
            return nullptr;
        return tg[0];
    }
    Operator userOfType(const Tensor &t, OpType type, size_t after) const {
        Operator u = onlyUser(t);
        if (!u || !(u->getOpType() == type) || pos(u) <= after)
            return nullptr;
        return u;
    }
    // memory no operator ever writes and the planner never recycles: graph weights and inputs (dataMalloc, graph.cc: weights
    // live in their own region, inputs / outputs are "not reused later"). A tensor the user created without marking it is
    // recycled after its last reader like any intermediate.
    static bool persistent(const Tensor &t) { return !t->getSource() && (t->isWeight() || t->isInput()); }
    static Tensor otherOf(const Operator &o, const Tensor &t) { return o->getInputs(0) == t ? o->getInputs(1) : o->getInputs(0); }
    static bool usesOnce(const Operator &o, const Tensor &t) {
        int c = 0;
        for (const auto &in : o->getInputs())
            c += in == t;
        return c == 1;
    }
    // `t` seen through Reshape-family operators down to a tensor no operator writes (a weight / graph input): a reshape
    // does not change the bytes, so the consumer can read the root at any time. Returns the root (or nullptr) and the
    // positions of the copy operators that become dead when this consumer was their chain's only reader.
    Tensor aliasRoot(const Tensor &t, std::vector<size_t> &dead) const {
        Tensor cur = t;
        std::vector<size_t> via;
        bool sole = true;
        while (true) {
            Operator src = cur->getSource();
            if (!src)
                break;
            if (!isCopyLike(src->getOpType()) || src->getInputs(0)->getBytes() != cur->getBytes())
                return nullptr;
            auto it = posOf.find(src.get());
            if (it == posOf.end())
                return nullptr;
            sole = sole && cur->getTargets().size() == 1 && !cur->isOutput() && !claimed[it->second];
            if (sole)
                via.push_back(it->second);
            cur = src->getInputs(0);
        }
        if (cur == t || !persistent(cur))
            return nullptr;
        dead = via;
        return cur;
    }
    // memory `t` occupies is not written by any operator at a position in (from, to) other than `members`
    // speculative: also trust the previous pass's `silent` set (forwarding checks only — the chains themselves are built
    // on facts, so that the second pass reproduces the first one's shapes wherever forwarding changes nothing)
    bool survives(const Tensor &t, size_t from, size_t to, const std::vector<size_t> &members, bool speculative = false) const {
        if (persistent(t) || from >= to)
            return true; // weights / graph inputs: no operator writes them, the planner never recycles them
        for (size_t p = from + 1; p < to; ++p) {
            if (std::find(members.begin(), members.end(), p) != members.end())
                continue;
            // an operator some item already owns writes when (and if) that item says so; an unclaimed one at its own place
            // (unless the previous pass found that it will be folded away: see `silent`)
            if (claimed[p] ? !(writeAt[p] > (long)from && writeAt[p] < (long)to) : (speculative && silent && (*silent)[p]))
                continue;
            for (const auto &o : ops[p]->getOutputs())
                if (o && overlaps(o, t))
                    return false;
            if (ops[p]->getOpType() == OpType::AttentionKVCache) // appends to its cache inputs in place
                for (int q = 0; q < 2; ++q)
                    if (overlaps(ops[p]->getInputs(q), t))
                        return false;
        }
        return true;
    }
    // nobody outside `members` reads `t` before position `to`
    bool unreadUntil(const Tensor &t, size_t to, const std::vector<size_t> &members) const {
        for (const auto &u : t->getTargets()) {
            auto it = posOf.find(u.get());
            if (it == posOf.end())
                return false;
            if (it->second < to && std::find(members.begin(), members.end(), it->second) == members.end())
                return false;
        }
        return true;
    }
    struct Read {
        Tensor t;
        size_t at; // position of the member that reads it in the unfused graph
    };
    bool readsSurvive(const std::vector<Read> &reads, size_t slot, const std::vector<size_t> &members) const {
        for (const auto &r : reads)
            if (!survives(r.t, r.at, slot, members))
                return false;
        return true;
    }
    void noteLateReads(const std::vector<Read> &reads, size_t slot) {
        for (const auto &r : reads)
            if (r.at < slot && !persistent(r.t))
                lateReads.push_back({r.at, slot, r.t});
    }
    // 16-byte alignment of device addresses; a dry run on another runtime's arena assumes the ROCM arena's 256-byte alignment
    bool al16(uintptr_t v) const { return !R || (v & 15) == 0; }
    int tunedVariant(const Operator &op) const {
        auto key = PerfEngine::Key{KernelAttrs{Device::ROCM, op->getOpType().underlying()}, op->getOpPerfKey()};
        auto rec = std::dynamic_pointer_cast<RocmVariantPerfRecordObj>(PerfEngine::getInstance().getPerfData(key));
        return rec ? rec->variant : -1;
    }
    bool mayUseWorkspace(int64_t b, int64_t m, int64_t nn) const {
        if (!rt)
            return false;
        int may = 1;
        ROCM_CALL(infini_rocm_matmul_may_use_workspace(rt, b, m, nn, &may));
        return may != 0;
    }
    // value of a one-element constant (a tensor no operator writes)
    // Weights only (the front-end's initializers): their values change through copyin / copy_inside alone, which drop the
    // scalar cache, the plans and the captured graphs. A graph INPUT may be rewritten behind the runtime's back (an externally
    // owned device pointer, a dlpack / torch write, another kernel): its value is never baked into a plan.
    bool scalarOf(const Tensor &t, double &v) const {
        if (t->size() != 1 || !persistent(t) || !t->isWeight() || !t->hasData())
            return false;
        const void *p = dataPtr(t);
        if (R) {
            auto it = R->scalarCache.find(p);
            if (it != R->scalarCache.end() && it->second.first == t->getBytes()) {
                v = it->second.second;
                return true;
            }
        }
        unsigned char buf[8] = {0};
        const size_t bytes = t->getBytes();
        if (bytes > 8)
            return false;
        t->getRuntime()->copyBlobToCPU(buf, p, bytes);
        const int dt = t->getDTypeIndex();
        if (dt == INFINI_DT_F32) {
            float f;
            std::memcpy(&f, buf, 4);
            v = f;
        } else if (dt == INFINI_DT_F64) {
            std::memcpy(&v, buf, 8);
        } else if (dt == INFINI_DT_F16) {
            uint16_t h;
            std::memcpy(&h, buf, 2);
            v = halfToDouble(h);
        } else if (dt == INFINI_DT_BF16) {
            uint32_t u = 0;
            std::memcpy(((char *)&u) + 2, buf, 2);
            float f;
            std::memcpy(&f, &u, 4);
            v = f;
        } else {
            return false;
        }
        if (R)
            R->scalarCache[p] = {bytes, v};
        return true;
    }
    bool scalarNear(const Tensor &t, double want, double rel) const {
        double v;
        return scalarOf(t, v) && std::fabs(v - want) <= rel * std::fabs(want);
    }

    // writers: the members whose own output tensors the item materialises (default: the last member)
    void emit(size_t slot, std::vector<size_t> members, std::string what, bool fused, std::function<void()> fn,
              std::vector<size_t> writers = {}) {
        for (size_t m : members) {
            claimed[m] = 1;
            writeAt[m] = -1;
        }
        if (writers.empty() && !members.empty())
            writers.push_back(*std::max_element(members.begin(), members.end()));
        for (size_t w : writers)
            writeAt[w] = (long)slot;
        if (log && fused) {
            std::string ms;
            for (size_t m : members)
                ms += (ms.empty() ? "" : ",") + std::to_string(m);
            fprintf(stderr, "[fusion] @%zu %s [%s]\n", slot, what.c_str(), ms.c_str());
        }
        PlanItem it;
        it.slot = slot;
        it.members = std::move(members);
        it.what = std::move(what);
        it.fused = fused;
        it.run = std::move(fn);
        items.push_back(std::move(it));
    }

    bool tryRules(size_t i) {
        const auto type = ops[i]->getOpType();
        if (type == OpType::MatMul)
            return planAttentionAt(i) || planMatmul(i) || planRowParallelAllReduce(i) || planGroupedAhead(i) || planIntoCopy(i);
        if (type == OpType::Transpose)
            return planAttentionFromTranspose(i) || planIntoCopy(i);
        if (type == OpType::Conv)
            return planConvStemPool(i) || planConv(i);
        if (type == OpType::Silu && planSiluMul(i))
            return true;
        if (type == OpType::Relu && planReluPool(i))
            return true;
        if (type == OpType::ReduceMean && planLayerNormDecomposed(i))
            return true;
        if ((type == OpType::Div || type == OpType::Mul) && planGeluDecomposed(i))
            return true;
        if (type == OpType::Add && (planAddNorm(i) || planBiasResidual(i) || planAddRelu(i)))
            return true;
        if (type == OpType::RoPE && planRopeHeadSplit(i, nullptr))
            return true;
        return planIntoCopy(i);
    }

#include "rocm_fusion_rules_decomposed.inc"
#include "rocm_fusion_rules_attention.inc"
#include "rocm_fusion_rules_matmul.inc"
#include "rocm_fusion_rules_conv.inc"
#include "rocm_fusion_rules_elementwise.inc"
};

namespace {
// Two passes when forwarding is in play: the first finds which operators get folded away, the second may let forwarded
// buffers outlive those (FusionPlanner's `silent`); its result is used only if its own folding confirms the assumption.
RocmRuntimeObj::LaunchPlan planTwice(const RocmRuntimeObj *R, const OpVec &ops, bool fusion) {
    static const bool twoPass = envOn("INFINI_ROCM_PLAN_TWO_PASS");
    std::vector<char> silent;
    RocmRuntimeObj::LaunchPlan first;
    {
        FusionPlanner p1(R, ops, fusion);
        first = p1.run();
        if (!fusion || !twoPass)
            return first;
        silent = p1.silentOps();
    }
    // iterate towards a self-consistent plan: each pass assumes the previous pass's silent set and is accepted as soon as
    // its own result confirms the assumption (a forwarding that succeeds changes who writes what: the head operator's own
    // buffer instead of the tail's)
    for (int pass = 2; pass <= 4; ++pass) {
        FusionPlanner pk(R, ops, fusion, &silent);
        RocmRuntimeObj::LaunchPlan next = pk.run();
        const bool holds = pk.silentHolds();
        if (std::getenv("INFINI_ROCM_FUSION_LOG"))
            fprintf(stderr, "[fusion] pass %d: %zu items (first pass %zu), %zu forwarded tensors, assumption %s\n", pass, next.items.size(),
                    first.items.size(), pk.forwardedTensors(), holds ? "holds" : "violated");
        if (holds)
            return next.items.size() <= first.items.size() ? next : first;
        silent = pk.silentOps();
    }
    return first;
}
} // namespace

RocmRuntimeObj::LaunchPlan RocmRuntimeObj::buildPlan(const Graph &graph) const {
    return planTwice(this, graph->getOperators(), fusion);
}

std::vector<std::string> RocmRuntimeObj::describeFusionPlan(const Graph &graph) {
    IT_ASSERT(graph != nullptr);
    graph->validateMemory();
    auto self = std::dynamic_pointer_cast<RocmRuntimeObj>(graph->getRuntime());
    static const bool envFusion = !(std::getenv("INFINI_ROCM_FUSION") && std::string(std::getenv("INFINI_ROCM_FUSION")) == "0");
    const auto plan = planTwice(self.get(), graph->getOperators(), self ? self->fusion : envFusion);
    std::vector<std::string> out;
    for (const auto &it : plan.items) {
        std::string s = std::to_string(it.slot) + " " + (it.fused ? it.what : std::string("op")) + " [";
        for (size_t q = 0; q < it.members.size(); ++q)
            s += (q ? "," : "") + std::to_string(it.members[q]);
        out.push_back(s + "]");
    }
    return out;
}

} // namespace infini
