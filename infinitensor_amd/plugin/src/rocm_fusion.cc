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
            return planAttentionAt(i) || planMatmul(i) || planRowParallelAllReduce(i) || planGroupedAhead(i);
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

    // ============================================================================================================
    // decomposed LayerNorm:  m = ReduceMean(x, last axis, keepdims); d = Sub(x, m); v = ReduceMean(Pow(d, 2) | Mul(d, d));
    //                        y = Div(d, Sqrt(Add(v, eps))) [* gamma] [+ beta]
    // ============================================================================================================
    struct NormMatch {
        std::vector<size_t> members; // ascending
        Tensor x, gamma, beta, out;
        double eps = 0;
        size_t first = 0, last = 0;
    };
    // `x` is read by exactly ReduceMean and Sub (the head of the pattern); extra readers of x are allowed when
    // `allowOtherReaders` (x stays a materialised tensor then)
    bool matchLayerNormDecomposed(const Tensor &x, NormMatch &m) const {
        static const bool on = envOn("INFINI_ROCM_FUSE_DECOMPOSED");
        if (!on)
            return false;
        Operator mean = nullptr, sub = nullptr;
        for (const auto &u : x->getTargets()) {
            auto it = posOf.find(u.get());
            if (it == posOf.end() || claimed[it->second])
                continue;
            if (u->getOpType() == OpType::ReduceMean && !mean)
                mean = u;
            else if (u->getOpType() == OpType::Sub && u->getInputs(0) == x && !sub)
                sub = u;
        }
        if (!mean || !sub)
            return false;
        const auto &xd = x->getDims();
        const int rank = xd.size();
        auto lastAxisMean = [&](const Operator &o) {
            auto r = as<ReduceBaseObj>(o);
            return r->getKeepDims() && r->getAxes().size() == 1 && *r->getAxes().begin() == rank - 1;
        };
        if (rank < 1 || !lastAxisMean(mean))
            return false;
        const Tensor mu = mean->getOutput(), d = sub->getOutput();
        if (sub->getInputs(1) != mu || mu->isOutput() || mu->getTargets().size() != 1 || d->isOutput() || d->getDims() != xd)
            return false;
        // d feeds the variance branch (Pow(d, 2) or Mul(d, d)) and the final Div
        Operator sq = nullptr, div = nullptr;
        const auto dt = d->getTargets();
        for (const auto &u : dt) {
            if (claimed[pos(u)])
                return false;
            if (u->getOpType() == OpType::Pow && u->getInputs(0) == d)
                sq = u;
            else if (u->getOpType() == OpType::Mul && u->getInputs(0) == d && u->getInputs(1) == d)
                sq = u;
            else if (u->getOpType() == OpType::Div && u->getInputs(0) == d)
                div = u;
            else
                return false;
        }
        if (!sq || !div || (sq->getOpType() == OpType::Pow ? dt.size() != 2 : dt.size() != 3))
            return false;
        if (sq->getOpType() == OpType::Pow && !scalarNear(sq->getInputs(1), 2.0, 1e-6))
            return false;
        Operator mean2 = userOfType(sq->getOutput(), OpType::ReduceMean, pos(sq));
        if (!mean2 || !lastAxisMean(mean2))
            return false;
        Operator addEps = userOfType(mean2->getOutput(), OpType::Add, pos(mean2));
        if (!addEps)
            return false;
        double eps;
        if (!scalarOf(otherOf(addEps, mean2->getOutput()), eps) || !(eps >= 0) || eps > 1e-2)
            return false;
        Operator sqrt = userOfType(addEps->getOutput(), OpType::Sqrt, pos(addEps));
        if (!sqrt || div->getInputs(1) != sqrt->getOutput() || sqrt->getOutput()->getTargets().size() != 1 ||
            sqrt->getOutput()->isOutput() || pos(div) < pos(sqrt))
            return false;
        m.members = {pos(mean), pos(sub), pos(sq), pos(mean2), pos(addEps), pos(sqrt), pos(div)};
        m.x = x;
        m.eps = eps;
        m.gamma = m.beta = nullptr;
        Tensor cur = div->getOutput();
        size_t lastPos = pos(div);
        const int nlast = xd.back();
        if (Operator mul = userOfType(cur, OpType::Mul, lastPos)) {
            const Tensor g = otherOf(mul, cur);
            if (g != cur && persistent(g) && isRowVector(g->getDims(), nlast) && g->getDType() == x->getDType() &&
                mul->getOutput()->getDims() == xd) {
                m.gamma = g;
                m.members.push_back(pos(mul));
                cur = mul->getOutput();
                lastPos = pos(mul);
                if (Operator add = userOfType(cur, OpType::Add, lastPos)) {
                    const Tensor b = otherOf(add, cur);
                    if (b != cur && persistent(b) && isRowVector(b->getDims(), nlast) && b->getDType() == x->getDType() &&
                        add->getOutput()->getDims() == xd) {
                        m.beta = b;
                        m.members.push_back(pos(add));
                        cur = add->getOutput();
                    }
                }
            }
        }
        if (!m.gamma)
            return false; // the kernels take a scale vector; a bare normalisation is not worth a special case
        std::sort(m.members.begin(), m.members.end());
        m.first = m.members.front();
        m.last = m.members.back();
        m.out = cur;
        if (!(cur->getDType() == x->getDType()))
            return false;
        return true;
    }

    bool planLayerNormDecomposed(size_t i) {
        const Tensor x = ops[i]->getInputs(0);
        NormMatch m;
        if (!matchLayerNormDecomposed(x, m) || m.first != i)
            return false;
        if (!readsSurvive({{x, m.first}}, m.last, m.members))
            return false;
        if (overlaps(m.out, x) && !samePlace(m.out, x))
            return false;
        if (overlaps(m.out, m.gamma) || (m.beta && overlaps(m.out, m.beta)))
            return false;
        noteLateReads({{x, m.first}}, m.last);
        const RocmRuntimeObj *r = R;
        const Tensor xx = x, g = m.gamma, b = m.beta, out = m.out;
        const int64_t nn = x->getDims().back(), outer = (int64_t)x->size() / nn;
        const float eps = (float)m.eps;
        emit(m.last, m.members, "layer_norm(decomposed)", true, [r, xx, g, b, out, nn, outer, eps] {
            ROCM_CALL(infini_rocm_layer_norm(r->handle(), xx->getDTypeIndex(), dataPtr(xx), dataPtr(g),
                                             b ? dataPtr(b) : nullptr, dataPtr(out), outer, nn,
                                             (int64_t)g->size(), b ? (int64_t)b->size() : 0, eps));
        });
        return true;
    }

    // ============================================================================================================
    // decomposed Gelu:  y = 0.5 * x * (1 + erf(x / sqrt 2)) as  Div(x, 1.41421) | Mul(x, 0.70711) -> Erf -> Add(1) ->
    //                   Mul(x, .) -> Mul(., 0.5)      (or Mul(x, 0.5) first, then Mul with the (1 + erf) branch)
    // ============================================================================================================
    struct GeluMatch {
        std::vector<size_t> members;
        Tensor x, out;
        size_t first = 0, last = 0;
    };
    bool matchGeluDecomposed(const Tensor &x, GeluMatch &m) const {
        static const bool on = envOn("INFINI_ROCM_FUSE_DECOMPOSED");
        if (!on)
            return false;
        const auto tg = x->getTargets();
        if (tg.size() != 2 || x->isOutput())
            return false;
        for (int a = 0; a < 2; ++a) {
            const Operator scale = tg[a], other = tg[1 - a];
            if (claimed[pos(scale)] || claimed[pos(other)])
                return false;
            // scale: x / sqrt(2) or x * (1 / sqrt(2))
            bool isScale = false;
            if (scale->getOpType() == OpType::Div && scale->getInputs(0) == x)
                isScale = scalarNear(scale->getInputs(1), 1.4142135623730951, 2e-3);
            else if (scale->getOpType() == OpType::Mul && usesOnce(scale, x))
                isScale = scalarNear(otherOf(scale, x), 0.7071067811865476, 2e-3);
            if (!isScale)
                continue;
            Operator erf = userOfType(scale->getOutput(), OpType::Erf, pos(scale));
            if (!erf)
                continue;
            Operator add1 = userOfType(erf->getOutput(), OpType::Add, pos(erf));
            if (!add1 || !scalarNear(otherOf(add1, erf->getOutput()), 1.0, 1e-6))
                continue;
            const Tensor onePlus = add1->getOutput();
            // x's second reader multiplies x with (1 + erf) or with 0.5
            if (!(other->getOpType() == OpType::Mul) || !usesOnce(other, x))
                continue;
            const Tensor o2 = otherOf(other, x);
            Operator final = nullptr;
            if (o2 == onePlus && onlyUser(onePlus) == other && pos(other) > pos(add1)) {
                // Mul(x, 1 + erf) -> Mul(., 0.5)
                Operator half = userOfType(other->getOutput(), OpType::Mul, pos(other));
                if (!half || !scalarNear(otherOf(half, other->getOutput()), 0.5, 1e-6))
                    continue;
                final = half;
                m.members = {pos(scale), pos(erf), pos(add1), pos(other), pos(half)};
            } else if (scalarNear(o2, 0.5, 1e-6)) {
                // Mul(x, 0.5) -> Mul(., 1 + erf)
                Operator prod = userOfType(other->getOutput(), OpType::Mul, pos(other));
                if (!prod || otherOf(prod, other->getOutput()) != onePlus || onlyUser(onePlus) != prod || pos(prod) < pos(add1))
                    continue;
                final = prod;
                m.members = {pos(scale), pos(erf), pos(add1), pos(other), pos(prod)};
            } else {
                continue;
            }
            const Tensor out = final->getOutput();
            if (out->getDims() != x->getDims() || !(out->getDType() == x->getDType()))
                continue;
            std::sort(m.members.begin(), m.members.end());
            m.first = m.members.front();
            m.last = m.members.back();
            m.x = x;
            m.out = out;
            return true;
        }
        return false;
    }
    bool planGeluDecomposed(size_t i) {
        for (const auto &x : ops[i]->getInputs()) {
            GeluMatch m;
            if (x->size() <= 1 || !matchGeluDecomposed(x, m) || m.first != i)
                continue;
            if (!readsSurvive({{x, m.first}}, m.last, m.members) || (overlaps(m.out, x) && !samePlace(m.out, x)))
                continue;
            noteLateReads({{x, m.first}}, m.last);
            const RocmRuntimeObj *r = R;
            const Tensor xx = x, out = m.out;
            emit(m.last, m.members, "gelu(decomposed)", true, [r, xx, out] {
                ROCM_CALL(infini_rocm_unary(r->handle(), INFINI_UN_GELU, xx->getDTypeIndex(), dataPtr(xx),
                                            dataPtr(out), out->size(), NAN, NAN));
            });
            return true;
        }
        return false;
    }

    // ============================================================================================================
    // attention
    // ============================================================================================================
    struct AttnPlan {
        std::vector<size_t> members;
        std::vector<Read> reads;
        Tensor q, kbuf, v, mask, scale, dstT, out;
        int b = 0, h = 0, sq = 0, sk = 0, d = 0, dt = 0;
        bool isDiv = false, mask2d = false;
        int64_t heads = 0;
        size_t last = 0;
    };
    // mm1 at `i`. `kAs`: a tensor whose BUFFER will hold K as [b, h, Sk, D] when the attention runs although the graph
    // says otherwise (its producer stores K head-major on our behalf), or nullptr. `absorbed`: a Transpose(0, 1, 3, 2)
    // in front of mm1 that is not launched (position, or SIZE_MAX).
    bool matchAttention(size_t i, const Tensor &kAs, size_t absorbed, AttnPlan &a) const {
        static const bool on = envOn("INFINI_ROCM_FUSE_ATTENTION");
        if (!on || claimed[i])
            return false;
        auto mm1 = as<MatmulObj>(ops[i]);
        if (mm1->getTransA() || mm1->getBias() || mm1->getAct() != ActType::None)
            return false;
        const Tensor q = mm1->getInputs(0), kx = mm1->getInputs(1);
        Tensor kbuf;
        Shape kd;
        size_t kReadAt = i;
        if (mm1->getTransB()) {
            kbuf = kx;
            kd = kx->getDims();
        } else if (kAs && kAs == kx) {
            kbuf = kx;
            const auto &x = kx->getDims(); // [b, h, D, Sk] as the graph sees it
            if (x.size() != 4)
                return false;
            kd = {x[0], x[1], x[3], x[2]};
        } else if (absorbed != SIZE_MAX) {
            kbuf = ops[absorbed]->getInputs(0);
            kd = kbuf->getDims();
            kReadAt = absorbed;
            if (ops[absorbed]->getOutput() != kx)
                return false;
        } else {
            return false;
        }
        const auto &qd = q->getDims();
        const int dt = q->getDTypeIndex();
        if (qd.size() != 4 || kd.size() != 4 || qd[0] != kd[0] || qd[1] != kd[1] || qd[3] != kd[3] || (qd[3] != 64 && qd[3] != 128) ||
            !isHalf(dt) || !(kx->getDType() == q->getDType()))
            return false;
        const int b = qd[0], h = qd[1], sq = qd[2], sk = kd[2], d = qd[3];
        // limits of infini_rocm_attention_ex (attention.hip): outside them the chain simply runs unfused
        if ((int64_t)b * h >= 65536 || h >= 65536 || sk <= 0 || sq <= 0 || !al16(addrOf(q) | addrOf(kbuf)))
            return false;
        a = AttnPlan();
        a.members = {i};
        if (absorbed != SIZE_MAX)
            a.members.push_back(absorbed);
        a.reads = {{q, i}, {kbuf, kReadAt}};
        Tensor cur = mm1->getOutput();
        size_t at = i;
        auto step = [&](const Operator &u) {
            a.members.push_back(pos(u));
            at = pos(u);
            cur = u->getOutput();
        };
        Operator u = onlyUser(cur);
        if (u && pos(u) > at && (u->getOpType() == OpType::Div || u->getOpType() == OpType::Mul)) {
            const Tensor a0 = u->getInputs(0), a1 = u->getInputs(1);
            a.isDiv = u->getOpType() == OpType::Div;
            const Tensor other = a0 == cur ? a1 : a0;
            if (other == cur || other->size() != 1 || !(other->getDType() == q->getDType()) || (a.isDiv && a0 != cur))
                return false;
            a.scale = other;
            a.reads.push_back({other, pos(u)});
            step(u);
            u = onlyUser(cur);
        }
        if (u && pos(u) > at && u->getOpType() == OpType::Add) {
            const Tensor other = otherOf(u, cur);
            const auto &md = other->getDims();
            // key mask [b|1, 1, 1, Sk] (BERT padding) or a full additive mask [b|1, h|1, Sq, Sk] with the same grouping rule:
            // heads may only broadcast when the batch does too or both are explicit ([1,1], [b,1], [b,h])
            const bool keyMask = md.size() == 4 && md[1] == 1 && md[2] == 1;
            const bool fullMask = md.size() == 4 && md[2] == sq && sq > 1 && (md[1] == 1 || (md[1] == h && md[0] == b));
            if (other == cur || md.size() != 4 || (md[0] != b && md[0] != 1) || !(keyMask || fullMask) || md[3] != sk ||
                !(other->getDType() == q->getDType()))
                return false;
            a.mask = other;
            a.mask2d = !keyMask;
            a.reads.push_back({other, pos(u)});
            step(u);
            u = onlyUser(cur);
        }
        if (!u || pos(u) <= at || !(u->getOpType() == OpType::Softmax) || as<SoftmaxObj>(u)->getAxis() != 3)
            return false;
        step(u);
        u = onlyUser(cur);
        if (!u || pos(u) <= at || !(u->getOpType() == OpType::MatMul))
            return false;
        auto mm2 = as<MatmulObj>(u);
        const Tensor v = mm2->getInputs(1), out = mm2->getOutput();
        if (mm2->getInputs(0) != cur || mm2->getTransA() || mm2->getTransB() || mm2->getBias() || mm2->getAct() != ActType::None ||
            v->getDims() != kd || !(v->getDType() == q->getDType()) || !al16(addrOf(v)) || !al16(addrOf(out) << 1))
            return false;
        a.reads.push_back({v, pos(u)});
        step(u);
        // Head merge: ctx [b, h, Sq, D] -> Transpose(0, 2, 1, 3) -> Reshape [b, Sq, h * D] (what every exported transformer
        // does before the output projection) is folded into the kernel's store (infini_rocm_attention_ex).
        a.dstT = out;
        a.heads = 0;
        static const bool mergeOn = envOn("INFINI_ROCM_FUSE_HEADMERGE");
        if (mergeOn) {
            Operator tr = userOfType(out, OpType::Transpose, at);
            if (tr && permIs(as<TransposeObj>(tr)->getPermute(), 0, 2, 1, 3)) {
                Operator rs = userOfType(tr->getOutput(), OpType::Reshape, pos(tr));
                if (rs && rs->getOutput()->getBytes() == out->getBytes() && rs->getOutput()->getDType() == out->getDType()) {
                    a.members.push_back(pos(tr));
                    a.members.push_back(pos(rs));
                    a.dstT = rs->getOutput();
                    a.heads = h;
                    at = pos(rs);
                }
            }
        }
        std::sort(a.members.begin(), a.members.end());
        a.last = a.members.back();
        a.q = q, a.kbuf = kbuf, a.v = v, a.out = out;
        a.b = b, a.h = h, a.sq = sq, a.sk = sk, a.d = d, a.dt = dt;
        if (!readsSurvive(a.reads, a.last, a.members))
            return false;
        return true;
    }

    struct MMChain;
    void commitAttention(const AttnPlan &a) {
        // O may sit exactly on Q (the planner likes to: Q is dead after the first MatMul and has O's size): a workgroup
        // loads its query rows before the key sweep and writes the same rows of O after it (plain layout only). K / V are
        // read by everyone. Any other overlap (K and V die after their MatMul too, and have O's size) is bridged through the
        // workspace: O is [Sq, D] per head, the copy is small next to the score traffic the fusion removes.
        const bool onQ = a.heads == 0 && addrOf(a.dstT) == addrOf(a.q) && a.dstT->getDims() == a.q->getDims();
        const bool hazard = (overlaps(a.dstT, a.q) && !onQ) || overlaps(a.dstT, a.kbuf) || overlaps(a.dstT, a.v) ||
                            (a.mask && overlaps(a.dstT, a.mask)) || (a.scale && overlaps(a.dstT, a.scale));
        // pairs (batch, head) served by one mask slab: [1,1,..] all of them, [b,1,..] the heads of a batch, [b,h,..] one
        const int64_t group = !a.mask ? 1 : ((a.mask->getDims()[1] == a.h && a.h > 1) ? 1 : (a.mask->getDims()[0] == 1 ? (int64_t)a.b * a.h : a.h));
        noteLateReads(a.reads, a.last);
        const RocmRuntimeObj *r = R;
        const AttnPlan ap = a;
        auto attn = [r, ap, group](void *dst) {
            ROCM_CALL(infini_rocm_attention_ex(r->handle(), ap.dt, dataPtr(ap.q), dataPtr(ap.kbuf),
                                               dataPtr(ap.v), ap.mask ? dataPtr(ap.mask) : nullptr, dst,
                                               (int64_t)ap.b * ap.h, ap.sq, ap.sk, ap.d, group,
                                               ap.scale ? dataPtr(ap.scale) : nullptr, ap.isDiv ? 1 : 0, 1.0f, 0, ap.heads,
                                               ap.mask2d ? 1 : 0));
        };
        if (!hazard) {
            emit(a.last, a.members, "attention", true, [attn, ap] { attn(dataPtr(ap.dstT)); });
            return;
        }
        // The bridged result usually feeds exactly one operator, the output projection, right behind the chain: let that
        // MatMul (with whatever the MatMul rule folds into it: bias, ...) read its A operand straight from the workspace
        // instead of copying 25 MB per BERT layer to a tensor nobody else reads. Only a MatMul that cannot itself take
        // the workspace (split-K partial planes).
        static const bool feedOn = envOn("INFINI_ROCM_FEED_NEXT");
        const size_t bytes = a.dstT->getBytes();
        // claim the attention first so that the consumer's chain cannot pick its members
        std::vector<size_t> members = a.members;
        for (size_t m : members)
            claimed[m] = 1;
        Operator nx = feedOn ? onlyUser(a.dstT) : nullptr;
        if (nx && nx->getOpType() == OpType::MatMul && pos(nx) == a.last + 1 &&
            !(R && R->comm && R->comm->getWorldSize() > 1 && userOfType(nx->getOutput(), OpType::AllReduceSum, pos(nx)))) {
            auto mmn = as<MatmulObj>(nx);
            const auto [nb, nm, nn, nk] = mmn->getBMNK();
            bool onlyA = mmn->getInputs(0) == a.dstT;
            for (size_t q = 1; q < mmn->getInputs().size(); ++q)
                onlyA = onlyA && mmn->getInputs(q) != a.dstT;
            if (!mayUseWorkspace(nb, nm, nn) && onlyA && tunedVariant(nx) != 3) {
                MMChain ch = buildMatmulChain(pos(nx), /*contiguousOnly*/ true, a.dstT);
                if (ch.ok && !ch.attn) {
                    if (ch.fwdTo) {
                        (*fwd)[ch.out.get()] = ch.fwdTo;
                        lateReads.push_back({ch.head, ch.fwdLastUse, ch.mm->getOutput()});
                    }
                    std::vector<size_t> all = members;
                    all.insert(all.end(), ch.members.begin(), ch.members.end());
                    noteLateReads(ch.reads, ch.slot);
                    auto chainRun = chainLaunch(ch, a.dstT.get(), bytes);
                    emit(ch.slot, all, "attention(bridged)>" + ch.what, true, [r, attn, chainRun, bytes] {
                        attn(r->getWorkspace(bytes));
                        chainRun();
                    }, ch.fwdTo ? std::vector<size_t>{ch.head} : std::vector<size_t>{});
                    return;
                }
            }
        }
        emit(a.last, members, "attention(bridged)+copy", true, [r, attn, ap, bytes] {
            void *ws = r->getWorkspace(bytes);
            attn(ws);
            ROCM_CALL(infini_rocm_copy_inside(r->handle(), dataPtr(ap.dstT), ws, bytes));
        });
    }

    bool planAttentionAt(size_t i) {
        AttnPlan a;
        if (!matchAttention(i, nullptr, SIZE_MAX, a))
            return false;
        commitAttention(a);
        return true;
    }
    // Transpose in front of Q.K^T. perm (0, 1, 3, 2) on the head-split K: absorbed, the kernel reads K itself.
    // perm (0, 2, 3, 1) on K's [B, S, H, D] view (two exporter transposes merged): run it as (0, 2, 1, 3) into the same
    // buffer — K head-major is what the kernel wants — unless K's producer already did (planMatmul handles that case).
    bool planAttentionFromTranspose(size_t i) {
        auto tr = as<TransposeObj>(ops[i]);
        const Tensor kx = tr->getOutput();
        Operator mm = onlyUser(kx);
        if (!mm || !(mm->getOpType() == OpType::MatMul) || pos(mm) <= i || mm->getInputs(1) != kx || mm->getInputs(0) == kx)
            return false;
        const auto perm = tr->getPermute();
        AttnPlan a;
        if (permIs(perm, 0, 1, 3, 2)) {
            if (!matchAttention(pos(mm), nullptr, i, a))
                return false;
            commitAttention(a);
            return true;
        }
        if (permIs(perm, 0, 2, 3, 1)) {
            if (!matchAttention(pos(mm), kx, SIZE_MAX, a))
                return false;
            const Tensor in = tr->getInputs(0);
            if (overlaps(in, kx))
                return false;
            const RocmRuntimeObj *r = R;
            emit(i, {i}, "transpose(K head-major)", true, [r, in, kx] {
                const auto &d = in->getDims();
                const int64_t shape[4] = {d[0], d[1], d[2], d[3]};
                const int p[4] = {0, 2, 1, 3};
                ROCM_CALL(infini_rocm_transpose(r->handle(), in->getDTypeIndex(), dataPtr(in), dataPtr(kx), 4,
                                                shape, p));
            });
            commitAttention(a);
            return true;
        }
        return false;
    }

    // ============================================================================================================
    // MatMul chains
    // ============================================================================================================
    struct MMChain {
        bool ok = false;
        std::shared_ptr<MatmulObj> mm;
        size_t head = 0, slot = 0;
        std::vector<size_t> members; // ascending, with the dead alias operators
        std::vector<Read> reads;
        Tensor bias;                 // folded Add operand (read through biasSrc)
        Tensor biasSrc;              // tensor whose buffer is passed (the alias root, or `bias` itself)
        int act = 0;                 // 5: Gelu in the epilogue
        int store = 0;               // 0 plain, 1 head split, 2 redirected into a Reshape-family copy's output
        bool kHeadMajor = false;     // store 1 into the buffer of a Transpose(0, 2, 3, 1) output (read by an attention)
        Tensor out;                  // tensor whose buffer receives the result
        long S = 0, D = 0, rows = 0;
        int n = 0, k = 0;
        std::optional<AttnPlan> attn; // committed together with the chain (kHeadMajor)
        std::string what;
        void *fwdTo = nullptr;        // the result goes to the MatMul's own output buffer and `out` is forwarded there
        size_t fwdLastUse = 0;
    };

    // The longest fusable chain headed by the MatMul at `i` (ok = false: nothing to fold — not even a redirect).
    // contiguousOnly: every member must directly follow the previous one (dead alias operators aside).
    // fedA: the MatMul's A operand is read from the workspace, not from its own buffer (the caller bridged it there), so the
    // chain's output may sit on that buffer.
    MMChain buildMatmulChain(size_t i, bool contiguousOnly, const Tensor &fedA = nullptr) {
        MMChain best;
        if (claimed[i] || !(ops[i]->getOpType() == OpType::MatMul))
            return best;
        auto mm = as<MatmulObj>(ops[i]);
        const auto [b, m, nn, kk] = mm->getBMNK();
        const Tensor A = mm->getInputs(0), W = mm->getInputs(1), C = mm->getOutput();
        const int dt = A->getDTypeIndex();
        MMChain c;
        c.mm = mm;
        c.head = i;
        c.members = {i};
        c.reads = {{W, i}};
        if (!(fedA && fedA == A))
            c.reads.push_back({A, i});
        if (mm->numInputs() == 3)
            c.reads.push_back({mm->getInputs(2), i});
        c.out = C;
        c.n = nn, c.k = kk;
        c.rows = (long)b * m;
        c.slot = i;
        c.what = "matmul";
        // a candidate is valid when the reads of its earlier members survive until its slot and its output buffer does
        // not overlap what the GEMM reads
        auto valid = [&](MMChain &x) {
            x.fwdTo = nullptr;
            if (const auto sfx = x.what.find(" (output forwarded)"); sfx != std::string::npos)
                x.what.erase(sfx, 19);
            if (contiguousOnly) {
                for (size_t q = 1; q < x.members.size(); ++q)
                    if (x.members[q] != x.members[q - 1] + 1)
                        return false;
            }
            if (!readsSurvive(x.reads, x.slot, x.members))
                return false;
            bool clash = x.biasSrc && overlaps(x.out, x.biasSrc);
            for (const auto &rd : x.reads)
                clash = clash || overlaps(x.out, rd.t);
            if (!clash)
                return true;
            // The planner put the chain's output on something the GEMM still reads (typically: the bias Add's output on the
            // MatMul's dead A operand). The MatMul's OWN output buffer was allocated while every operand was live: write
            // there and forward the chain's final tensor to it, if that block stays untouched until the tensor's last reader
            // (plain store forms only: a head-split / K-for-attention store has a consumer that addresses the buffer itself).
            static const bool fwdOn = envOn("INFINI_ROCM_FORWARD");
            if (!fwdOn || x.out == C || x.store == 1 || x.out->isOutput() || x.out->getBytes() != C->getBytes() ||
                x.out->getTargets().empty() || (x.biasSrc && overlaps(C, x.biasSrc)))
                return false;
            for (const auto &rd : x.reads)
                if (overlaps(C, rd.t))
                    return false;
            size_t lastUse = x.slot;
            for (const auto &u : x.out->getTargets()) {
                auto it = posOf.find(u.get());
                if (it == posOf.end() || it->second <= x.slot || claimed[it->second]) // (see planConv: an already-planned reader)
                    return false;
                lastUse = std::max(lastUse, it->second);
            }
            // from the MatMul's OWN position: the planner considers C free once its reader (the bias Add) ran, so a non-member
            // operator between the head and the slot may have been given C's block — the fused kernel's write at the slot
            // would clobber that operator's result (round-3 advisor finding)
            if (!survives(C, i, lastUse + 1, x.members, true))
                return false;
            x.fwdTo = C->getRawDataPtr<void *>();
            x.fwdLastUse = lastUse;
            x.what += " (output forwarded)";
            return true;
        };
        auto add = [&](MMChain &x, size_t p) {
            x.members.push_back(p);
            std::sort(x.members.begin(), x.members.end());
            x.slot = std::max(x.slot, p);
        };
        Tensor cur = C;
        size_t at = i;
        // 1. Add(row bias)
        static const bool biasOn = envOn("INFINI_ROCM_FUSE_MATMUL_BIAS");
        if (biasOn && mm->numInputs() == 2) {
            if (Operator u = userOfType(cur, OpType::Add, at)) {
                const Tensor other = otherOf(u, cur);
                std::vector<size_t> dead;
                if (other != cur && isRowVector(other->getDims(), nn) && other->getDType() == C->getDType() &&
                    u->getOutput()->getDims() == C->getDims()) {
                    MMChain x = c;
                    const Tensor root = aliasRoot(other, dead);
                    x.bias = other;
                    x.biasSrc = root ? root : other;
                    for (size_t dp : dead)
                        add(x, dp);
                    add(x, pos(u));
                    if (!root)
                        x.reads.push_back({other, pos(u)});
                    x.out = u->getOutput();
                    x.what += "+bias";
                    if (valid(x)) {
                        c = x;
                        best = c;
                        best.ok = true;
                        cur = c.out;
                        at = pos(u);
                    }
                }
            }
        }
        // 2. Gelu (single operator or the five-operator form) in the epilogue, f16 / bf16
        static const bool geluOn = envOn("INFINI_ROCM_FUSE_GELU");
        if (geluOn && isHalf(dt) && cur == c.out) {
            MMChain x = c;
            bool got = false;
            if (Operator u = userOfType(cur, OpType::Gelu, at)) {
                add(x, pos(u));
                x.out = u->getOutput();
                got = true;
            } else {
                GeluMatch g;
                if (matchGeluDecomposed(cur, g) && g.first > at) {
                    for (size_t p : g.members)
                        add(x, p);
                    x.out = g.out;
                    got = true;
                }
            }
            if (got && x.out->getBytes() == cur->getBytes() && x.out->getDType() == cur->getDType()) {
                x.act = 5;
                x.what += "+gelu";
                if (valid(x)) {
                    c = x;
                    best = c;
                    best.ok = true;
                    cur = c.out;
                    at = c.slot;
                }
            }
        }
        // 3. the store: head split, K head-major for an attention, or straight into a Reshape-family copy's output
        static const bool splitOn = envOn("INFINI_ROCM_FUSE_HEADSPLIT"), copyOn = envOn("INFINI_ROCM_FUSE_RESHAPE");
        if (Operator rs = onlyUser(cur); rs && pos(rs) > at && isCopyLike(rs->getOpType()) && rs->getInputs(0) == cur &&
                                        rs->getOutput()->getBytes() == cur->getBytes() && rs->getOutput()->getDType() == cur->getDType()) {
            const Tensor r = rs->getOutput();
            const auto &rd = r->getDims();
            bool split = false;
            if (splitOn && rs->getOpType() == OpType::Reshape && rd.size() == 4 && c.act == 0) {
                if (Operator tr = userOfType(r, OpType::Transpose, pos(rs))) {
                    const auto perm = as<TransposeObj>(tr)->getPermute();
                    const long B = rd[0], S = rd[1], Hh = rd[2], D = rd[3];
                    // the MatMul's rows are (batch, position), its columns (head, channel): [b x m] == [B x S] row-wise, n == H * D
                    const bool shapeOK = (long)b * m == B * S && (long)nn == Hh * D && m % S == 0 && D % 8 == 0 &&
                                         tr->getOutput()->getDType() == cur->getDType() && tr->getOutput()->getBytes() == cur->getBytes();
                    if (shapeOK && (permIs(perm, 0, 2, 1, 3) || permIs(perm, 0, 2, 3, 1))) {
                        MMChain x = c;
                        add(x, pos(rs));
                        add(x, pos(tr));
                        x.out = tr->getOutput();
                        x.store = 1;
                        x.S = S, x.D = D;
                        x.what += "+headsplit";
                        bool okx = true;
                        if (permIs(perm, 0, 2, 3, 1)) { // only as the K of a fused attention
                            okx = false;
                            Operator mm1 = onlyUser(x.out);
                            if (mm1 && mm1->getOpType() == OpType::MatMul && mm1->getInputs(1) == x.out && mm1->getInputs(0) != x.out &&
                                pos(mm1) > pos(tr)) {
                                // members of this chain must look claimed to the attention matcher
                                for (size_t p : x.members)
                                    claimed[p] = 1;
                                AttnPlan ap;
                                okx = matchAttention(pos(mm1), x.out, SIZE_MAX, ap);
                                for (size_t p : x.members)
                                    claimed[p] = 0;
                                if (okx) {
                                    x.attn = ap;
                                    x.kHeadMajor = true;
                                    x.what += "(K for attention)";
                                }
                            }
                        }
                        if (okx && valid(x)) {
                            c = x;
                            best = c;
                            best.ok = true;
                            split = true;
                        }
                    }
                }
            }
            if (!split && copyOn) {
                MMChain x = c;
                add(x, pos(rs));
                x.out = r;
                x.store = 2;
                x.what += ">reshape";
                if (valid(x)) {
                    c = x;
                    best = c;
                    best.ok = true;
                }
            }
        }
        if (!best.ok && contiguousOnly) { // the caller launches the bare MatMul through the chain machinery (input redirect)
            best = c;
            best.ok = true;
        }
        return best;
    }

    // the launch of one chain; feedT != nullptr: the MatMul reads `feedT` from the workspace (getWorkspace(feedBytes))
    std::function<void()> chainLaunch(const MMChain &c, const TensorObj *feedT, size_t feedBytes) const {
        const RocmRuntimeObj *r = R;
        const MMChain cc = c;
        const Operator op = ops[c.head];
        return [r, cc, op, feedT, feedBytes] {
            OverrideScope s;
            auto &o = RocmRuntimeObj::overrides;
            o.matmul = op.get();
            if (cc.biasSrc)
                o.biasPtr = dataPtr(cc.biasSrc);
            o.act = cc.act;
            if (cc.store == 1) {
                o.seq = (int)cc.S;
                o.headDim = (int)cc.D;
            }
            if (cc.fwdTo)
                ++r->forwardedCount;
            if (cc.out != cc.mm->getOutput()) // (a forwarded `out` resolves to the MatMul's own buffer: a no-op redirect)
                s.redirect(cc.mm->getOutput().get(), dataPtr(cc.out));
            if (feedT)
                s.redirect(feedT, r->getWorkspace(feedBytes));
            r->launchOne(op);
        };
    }

    void commitChain(const MMChain &c) {
        noteLateReads(c.reads, c.slot);
        if (c.fwdTo) {
            (*fwd)[c.out.get()] = c.fwdTo;
            lateReads.push_back({c.head, c.fwdLastUse, c.mm->getOutput()});
        }
        // (a forwarded chain writes the MatMul operator's own buffer, not its last member's)
        emit(c.slot, c.members, c.what, true, chainLaunch(c, nullptr, 0), c.fwdTo ? std::vector<size_t>{c.head} : std::vector<size_t>{});
        if (c.attn)
            commitAttention(*c.attn);
    }

    // MatMul at `i`: its chain, and — for head-split chains — the sibling projections of the same activation (q, k, v) as
    // ONE grouped launch when their weights / biases / outputs sit at uniform distances (the group index is the GEMM's
    // batch index with a zero A stride; BERT-base: 3 x 25 us -> ~62 us per layer). Same kernels, same sums: bit-identical.
    bool planMatmul(size_t i) {
        MMChain h0 = buildMatmulChain(i, false);
        if (!h0.ok)
            return false;
        static const bool groupOn = envOn("INFINI_ROCM_GROUP_QKV");
        const auto &mm0 = h0.mm;
        const Tensor a0 = mm0->getInputs(0), w0 = mm0->getInputs(1);
        const int dt = a0->getDTypeIndex();
        auto biasOf = [](const MMChain &c) -> Tensor { return c.biasSrc ? c.biasSrc : (c.mm->numInputs() == 3 ? c.mm->getInputs(2) : nullptr); };
        const Tensor bias0 = biasOf(h0);
        const bool groupable = groupOn && h0.store == 1 && h0.act == 0 && isHalf(dt) && !mm0->getTransA() && !mm0->getTransB() &&
                               w0->getRank() == 2 && tunedVariant(ops[i]) < 0 &&
                               (!bias0 || ((int)bias0->size() == h0.n && isRowVector(bias0->getDims(), h0.n)));
        if (groupable) {
            std::vector<MMChain> g{h0};
            // siblings: other unclaimed MatMuls reading a0 as their A operand, in operator order
            std::vector<size_t> sib;
            for (const auto &u : a0->getTargets()) {
                auto it = posOf.find(u.get());
                if (it == posOf.end() || it->second == i || claimed[it->second] || !(u->getOpType() == OpType::MatMul) ||
                    u->getInputs(0) != a0)
                    continue;
                sib.push_back(it->second);
            }
            std::sort(sib.begin(), sib.end());
            sib.erase(std::unique(sib.begin(), sib.end()), sib.end());
            // members of the group so far must look claimed while the next chain is built (the K chain's attention matcher)
            for (size_t j : sib) {
                if (g.size() >= 4)
                    break;
                if (j < i)
                    continue;
                MMChain hn = buildMatmulChain(j, false);
                if (!hn.ok || hn.store != 1 || hn.act != 0 || hn.fwdTo)
                    continue;
                const auto &mn = hn.mm;
                const Tensor wn = mn->getInputs(1), bn = biasOf(hn);
                if (mn->getTransA() || mn->getTransB() || wn->getDims() != w0->getDims() || !(wn->getDType() == w0->getDType()) ||
                    (bn != nullptr) != (bias0 != nullptr) || (bn && bn->size() != bias0->size()) || hn.S != h0.S || hn.D != h0.D ||
                    hn.rows != h0.rows || hn.n != h0.n || hn.k != h0.k || tunedVariant(ops[j]) >= 0)
                    continue;
                // chains must not share members (cannot, but an attention committed with one must not claim another's)
                g.push_back(hn);
            }
            if (g.size() >= 2 && commitGroup(g, a0))
                return true;
        }
        commitChain(h0);
        return true;
    }

    // One grouped launch for head-split projections g (same activation). The launch runs at the LAST member's slot; the
    // group index walks the members in the order of their OUTPUT addresses, so weights / biases must be uniformly spaced in
    // that same order. Members that cannot be arranged so are left out (they run on their own).
    bool commitGroup(std::vector<MMChain> g, const Tensor &a0) {
        auto biasOf = [](const MMChain &c) -> Tensor { return c.biasSrc ? c.biasSrc : (c.mm->numInputs() == 3 ? c.mm->getInputs(2) : nullptr); };
        std::sort(g.begin(), g.end(), [](const MMChain &x, const MMChain &y) { return addrOf(x.out) < addrOf(y.out); });
        const intptr_t es = (intptr_t)a0->getDType().getSize();
        const intptr_t cBytes = (intptr_t)g[0].out->getBytes();
        auto uniform = [&](const std::vector<MMChain> &s) {
            const intptr_t dw = (intptr_t)addrOf(s[1].mm->getInputs(1)) - (intptr_t)addrOf(s[0].mm->getInputs(1));
            const Tensor b0 = biasOf(s[0]);
            const intptr_t db = b0 ? (intptr_t)addrOf(biasOf(s[1])) - (intptr_t)addrOf(b0) : 0;
            if (dw % 16 != 0 || db % es != 0)
                return false;
            for (size_t j = 1; j < s.size(); ++j) {
                if ((intptr_t)addrOf(s[j].mm->getInputs(1)) - (intptr_t)addrOf(s[0].mm->getInputs(1)) != (intptr_t)j * dw)
                    return false;
                if (b0 && (intptr_t)addrOf(biasOf(s[j])) - (intptr_t)addrOf(b0) != (intptr_t)j * db)
                    return false;
                if ((intptr_t)addrOf(s[j].out) - (intptr_t)addrOf(s[0].out) != (intptr_t)j * cBytes)
                    return false;
            }
            return true;
        };
        // the longest run of address-adjacent members that is uniform
        std::vector<MMChain> pick;
        for (size_t lo = 0; lo < g.size() && pick.size() < 2; ++lo)
            for (size_t hi = g.size(); hi >= lo + 2; --hi) {
                std::vector<MMChain> s(g.begin() + lo, g.begin() + hi);
                if (uniform(s)) {
                    pick = s;
                    break;
                }
            }
        if (pick.size() < 2)
            return false;
        size_t slot = 0;
        std::vector<size_t> members;
        std::vector<Read> reads;
        for (const auto &c : pick) {
            slot = std::max(slot, c.slot);
            members.insert(members.end(), c.members.begin(), c.members.end());
            reads.insert(reads.end(), c.reads.begin(), c.reads.end());
        }
        std::sort(members.begin(), members.end());
        if (std::adjacent_find(members.begin(), members.end()) != members.end())
            return false;
        if (!readsSurvive(reads, slot, members))
            return false;
        for (const auto &c : pick) {
            // results written later than planned must have no reader in between; no output may land on an input
            if (c.slot < slot) {
                std::vector<size_t> allowed = members;
                if (c.attn)
                    allowed.insert(allowed.end(), c.attn->members.begin(), c.attn->members.end());
                if (!unreadUntil(c.out, slot + 1, allowed))
                    return false;
                // an attention committed with this member must run after the group
                if (c.attn && c.attn->last < slot)
                    return false;
            }
            for (const auto &rd : reads)
                if (overlaps(c.out, rd.t))
                    return false;
            if (Tensor bs = biasOf(c); bs && overlaps(c.out, bs))
                return false;
        }
        // a member whose output buffer is still in use by someone else before `slot`? No: the buffer is the member's own
        // from its planned position on; only members planned AFTER... all members' slots are <= slot, so every output
        // buffer is live at `slot`.
        noteLateReads(reads, slot);
        const RocmRuntimeObj *r = R;
        const Tensor w0 = pick[0].mm->getInputs(1), bias0 = biasOf(pick[0]), out0 = pick[0].out;
        const int64_t strideB = ((intptr_t)addrOf(pick[1].mm->getInputs(1)) - (intptr_t)addrOf(w0)) / es;
        const int64_t strideBias = bias0 ? ((intptr_t)addrOf(biasOf(pick[1])) - (intptr_t)addrOf(bias0)) / es : 0;
        const int64_t cnt = pick.size(), rows = pick[0].rows, S = pick[0].S, D = pick[0].D;
        const int nn = pick[0].n, kk = pick[0].k, dt = a0->getDTypeIndex();
        const Tensor A = a0;
        std::vector<size_t> writers;
        for (const auto &c : pick)
            writers.push_back(c.members.back());
        emit(slot, members, pick[0].what + " x" + std::to_string(cnt) + " (grouped)", true,
             [r, A, w0, bias0, out0, cnt, rows, nn, kk, strideB, strideBias, S, D, dt] {
                 ROCM_CALL(infini_rocm_matmul_headsplit(r->handle(), dt, dataPtr(A), dataPtr(w0),
                                                        bias0 ? dataPtr(bias0) : nullptr, dataPtr(out0), cnt,
                                                        rows, nn, kk, 0, 0, /*strideA*/ 0, strideB, strideBias, 0, bias0 ? 1 : 0, 0, S, D));
             },
             writers);
        for (const auto &c : pick)
            if (c.attn)
                commitAttention(*c.attn);
        // members left out run on their own
        for (const auto &c : g) {
            bool in = false;
            for (const auto &p : pick)
                in = in || p.head == c.head;
            if (!in && !claimed[c.head]) {
                bool free = true;
                for (size_t m : c.members)
                    free = free && !claimed[m];
                if (free && (!c.attn || !claimed[c.attn->members[0]]))
                    commitChain(c);
            }
        }
        return true;
    }

    // Plain MatMuls that multiply the SAME activations by different weights (a decoder block's gate and up projections; q and k
    // ahead of their RoPE) are a few operators apart in the list: mm_g, Silu, mm_u, Mul. Each alone leaves the chip part empty
    // (Llama-7B at 2048 tokens: 128 or 344 tiles of 256^2 on 256 CUs); as ONE grouped launch (infini_rocm_matmul_grouped: batch
    // index = member, zero A stride, the members' weights / outputs at their own uniform distances) they fill it: q + k + v
    // 215 -> 169 us, gate + up 330 -> 286 us through the C ABI. A later member runs EARLIER than its place in the list, so:
    //   * none of the operators it jumps over may produce (or overwrite) anything it reads;
    //   * its output buffer — which the planner handed out for the member's own position — must not overlap anything those
    //     operators (or the group, or a sunk item's late reads) read or write;
    //   * a member that another rule wants (head split, Gelu epilogue, bias fold, copy elision) is left to that rule.
    // INFINI_ROCM_GROUP_MATMUL=0 switches it off.
    bool planGroupedAhead(size_t i) {
        static const bool enabled = envOn("INFINI_ROCM_GROUP_MATMUL");
        if (!enabled)
            return false;
        auto eligible = [&](size_t j) -> bool {
            if (!(ops[j]->getOpType() == OpType::MatMul) || claimed[j] || tunedVariant(ops[j]) >= 0)
                return false;
            auto mm = as<MatmulObj>(ops[j]);
            const Tensor a = mm->getInputs(0), w = mm->getInputs(1);
            const int dt = a->getDTypeIndex();
            if (!isHalf(dt) || mm->getTransA() || w->getRank() != 2 || !(w->getDType() == a->getDType()))
                return false;
            if (mm->numInputs() == 3 && !(mm->getInputs(2)->getRank() == 1 && (int)mm->getInputs(2)->size() == w->getDims()[mm->getTransB() ? 0 : 1]))
                return false;
            MMChain other = buildMatmulChain(j, false); // someone else's pattern
            return !other.ok;
        };
        if (!eligible(i))
            return false;
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
                   t == OpType::RoPE || isCopyLike(t) || t == OpType::Transpose;
        };
        constexpr size_t kWindow = 12;
        for (size_t j = i + 1; j < n && j <= i + kWindow && members.size() < 4; ++j) {
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
                for (const auto &lr : lateReads) // a sunk item still reads this memory later than the graph says
                    if (member && clear && lr.to > i && lr.from < j && overlaps(outj, lr.t))
                        clear = false;
                for (size_t q = 0; member && q < producedBetween.size(); ++q)
                    for (const auto &in : mj->getInputs())
                        member = member && !overlaps(producedBetween[q], in);
                // The planner usually recycles: mm_u's output sits where mm_g's was (dead once Silu has read it). Then the member's
                // result is PARKED in the workspace and its one consumer — the very next operator, an element-wise / RoPE kernel —
                // reads it from there. Two-member groups only (member = batch index needs ONE output stride), nothing
                // between the group and that consumer may use the workspace, and the grouped MatMul itself must not (split-K).
                if (member && !clear) {
                    const auto [gb, gm, gn, gk] = mj->getBMNK();
                    bool quiet = members.size() == 1 && parkAt == 0 && !mayUseWorkspace(2, (int64_t)gb * gm, gn) && j + 1 < n &&
                                 onlyUser(outj) == ops[j + 1] && wsQuiet(ops[j + 1]) && tunedVariant(ops[j + 1]) < 0;
                    for (size_t bq = i + 1; quiet && bq < j; ++bq)
                        quiet = wsQuiet(ops[bq]) && !claimed[bq];
                    if (quiet && usesOnce(ops[j + 1], outj))
                        parkAt = j;
                    else
                        member = false;
                }
            }
            if (claimed[j] && !member) // an operator another item already owns: its real run time is not its position
                break;
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
        auto addr = [](const Tensor &t) { return (intptr_t)addrOf(t); };
        const intptr_t es = (intptr_t)a0->getDType().getSize();
        const intptr_t cBytes = (intptr_t)mm0->getOutput()->getBytes();
        auto Wt = [&](size_t q) { return as<MatmulObj>(ops[members[q]])->getInputs(1); };
        auto Bi = [&](size_t q) { return as<MatmulObj>(ops[members[q]])->getInputs(2); };
        auto Ot = [&](size_t q) { return ops[members[q]]->getOutput(); };
        auto uniform = [&](size_t cnt, intptr_t &dw, intptr_t &db, intptr_t &dc) {
            dw = addr(Wt(1)) - addr(w0);
            db = bias0 ? addr(Bi(1)) - addr(bias0) : 0;
            dc = addr(Ot(1)) - addr(Ot(0));
            if (dw % 16 != 0 || db % es != 0 || dc % 16 != 0 || (dc < cBytes && dc > -cBytes))
                return false;
            for (size_t q = 2; q < cnt; ++q)
                if (addr(Wt(q)) - addr(w0) != (intptr_t)q * dw || (bias0 && addr(Bi(q)) - addr(bias0) != (intptr_t)q * db) ||
                    addr(Ot(q)) - addr(Ot(0)) != (intptr_t)q * dc)
                    return false;
            return true;
        };
        size_t cnt = members.size();
        if (cnt < 2)
            return false;
        intptr_t dw = 0, db = 0, dc = 0;
        if (parkAt) { // exactly two members: the second one's result goes to the workspace
            dw = addr(Wt(1)) - addr(w0);
            db = bias0 ? addr(Bi(1)) - addr(bias0) : 0;
            if (cnt != 2 || dw % 16 != 0 || db % es != 0)
                return false;
        } else {
            while (cnt >= 2 && !uniform(cnt, dw, db, dc))
                --cnt;
            if (cnt < 2)
                return false;
        }
        // rows: the batch folds into m when the weight is shared (rank-2 w) and A is dense
        const int64_t rows = (int64_t)b0 * m0;
        const RocmRuntimeObj *r = R;
        const Tensor A = a0, W = w0, Bs = bias0, O = mm0->getOutput();
        const int tb = mm0->getTransB() ? 1 : 0, dt = a0->getDTypeIndex();
        const int64_t n64 = n0, k64 = k0;
        const bool park = parkAt != 0;
        std::vector<size_t> mem(members.begin(), members.begin() + cnt);
        emit(i, mem, "matmul x" + std::to_string(cnt) + " (grouped ahead" + (park ? ", parked)" : ")"), true,
             [r, A, W, Bs, O, cnt, rows, n64, k64, tb, dw, db, dc, es, dt, park, cBytes] {
                 intptr_t dcv = dc;
                 if (park) {
                     void *p = r->getWorkspace((size_t)cBytes);
                     dcv = (intptr_t)p - (intptr_t)dataPtr(O);
                     IT_ASSERT(dcv % 16 == 0 && !(dcv < cBytes && dcv > -cBytes), "parked GEMM output collides with the group's own");
                     ++r->parkedCount;
                 }
                 ROCM_CALL(infini_rocm_matmul_grouped(r->handle(), dt, dataPtr(A), dataPtr(W),
                                                      Bs ? dataPtr(Bs) : nullptr, dataPtr(O), (int64_t)cnt,
                                                      rows, n64, k64, 0, tb, /*strideA*/ 0, dw / es, dcv / es, db / es, 0, Bs ? 1 : 0, 0, 0, 0));
             },
             park ? std::vector<size_t>{i} : mem);
        if (parkAt)
            parked[parkAt + 1] = ParkedFeed{ops[parkAt]->getOutput().get(), (size_t)cBytes};
        return true;
    }

    // Row-parallel MatMul -> AllReduceSum (the o_proj / down projections of a tensor-parallel block, parallel_opt.py:195-210):
    // the reference issues ONE whole-tensor ncclAllReduce behind the whole GEMM (all_reduce.cc:10-33), so the xGMI links idle
    // during the GEMM and the matrix cores during the exchange. Here the GEMM is cut into row chunks (tokens); each chunk's
    // all-reduce goes to the runtime's comm stream as soon as its GEMM is enqueued and runs under the next chunk's GEMM; the
    // runtime stream joins once at the end (infini_rocm_all_reduce_async / comm_join; capturable). Same operands per output
    // element; a chunk's GEMM may pick another tile form than the whole GEMM (fp16 rounding of the summation order). Only
    // with more than one rank (INFINI_ROCM_TP_OVERLAP=0 off, =force also at world 1: tests).
    bool planRowParallelAllReduce(size_t i) {
        static const char *env = std::getenv("INFINI_ROCM_TP_OVERLAP");
        static const int mode = !env ? 1 : (std::string(env) == "force" ? 2 : std::atoi(env));
        if (mode == 0 || !R || !R->comm || (mode != 2 && R->comm->getWorldSize() < 2))
            return false;
        auto mm = as<MatmulObj>(ops[i]);
        const Tensor A = mm->getInputs(0), W = mm->getInputs(1), C = mm->getOutput();
        Operator ar = userOfType(C, OpType::AllReduceSum, i);
        // Rank symmetry: whether this rule fires decides how many collectives (4 chunked / 1 whole-tensor) a rank issues on the
        // shared communicator, so it may depend on the graph and on INFINI_ROCM_TP_OVERLAP only (which must be set alike on
        // every rank) — never on per-process state such as PerfEngine records of a tune() only some ranks ran.
        if (!ar || mm->numInputs() != 2 || mm->getTransA() || W->getRank() != 2)
            return false;
        const auto [b, m, nn, kk] = mm->getBMNK();
        const int64_t rows = (int64_t)b * m;
        constexpr int kChunks = 4;
        if (rows % kChunks != 0 || rows / kChunks < 256)
            return false;
        const Tensor Y = ar->getOutput();
        if (Y->getBytes() != C->getBytes() || overlaps(Y, A) || overlaps(Y, W) || overlaps(C, A) || overlaps(C, W))
            return false;
        std::vector<size_t> members{i, pos(ar)};
        std::vector<Read> reads{{A, i}, {W, i}};
        if (!readsSurvive(reads, pos(ar), members))
            return false;
        noteLateReads(reads, pos(ar));
        const RocmRuntimeObj *r = R;
        const int dt = A->getDTypeIndex(), tb = mm->getTransB() ? 1 : 0;
        const int64_t n64 = nn, k64 = kk, es = (int64_t)A->getDType().getSize();
        emit(pos(ar), members, "matmul>allreduce (4 row chunks, overlapped)", true, [r, A, W, C, Y, rows, n64, k64, es, dt, tb] {
            const int64_t rc = rows / kChunks;
            for (int c = 0; c < kChunks; ++c) {
                const char *a = (const char *)dataPtr(A) + (size_t)c * rc * k64 * es;
                char *cc = (char *)dataPtr(C) + (size_t)c * rc * n64 * es;
                char *y = (char *)dataPtr(Y) + (size_t)c * rc * n64 * es;
                ROCM_CALL(infini_rocm_matmul(r->handle(), dt, a, dataPtr(W), nullptr, cc, 1, rc, n64, k64, 0, tb, 0, 0, 0, 0, 0, 0));
                ROCM_CALL(infini_rocm_all_reduce_async(r->handle(), 0, dt, cc, y, rc * n64));
            }
            ROCM_CALL(infini_rocm_comm_join(r->handle()));
        });
        return true;
    }

    // the operator at `i` reads a parked group result: rules that know how to (RoPE head split, Silu-Mul reach it themselves)
    bool planParkedConsumer(size_t i, const ParkedFeed &pf) {
        if (ops[i]->getOpType() == OpType::RoPE)
            return planRopeHeadSplit(i, &pf);
        return false;
    }

    // ============================================================================================================
    // Conv chains
    // ============================================================================================================
    // Conv(7 x 7 / 2, C = 3 -> F = 64) -> Reshape(bias) -> Add -> Relu -> MaxPool(3 x 3 / 2 / 1): the stem of a CNN as ONE launch
    // (csrc/conv_stem.hip): the conv tile is pooled out of LDS, the 205 MB (batch 128) the conv used to write and the pool to
    // re-read never exist. Only the exact chain the library serves (infini_rocm_conv2d_pool_supported); every intermediate has
    // one reader; the pooled output must not sit on anything the conv reads.
    bool planConvStemPool(size_t i) {
        static const bool on = envOn("INFINI_ROCM_FUSE_STEM_POOL");
        if (!on)
            return false;
        auto conv = as<ConvObj>(ops[i]);
        const Tensor x = conv->getInputs(0), w = conv->getInputs(1), y = conv->getOutput();
        if (y->getDims().size() != 4)
            return false;
        const auto [nb, ch, hh, wd, ff, rr, ss] = conv->getNCHWFRS();
        const auto [ph, pw, sh, sw, dh, dw] = conv->getPadStrideDilation();
        Operator add = userOfType(y, OpType::Add, i);
        if (!add)
            return false;
        const Tensor other = otherOf(add, y);
        if (other == y || !isChannelBias(other->getDims(), ff) || !(other->getDType() == y->getDType()) ||
            add->getOutput()->getDims() != y->getDims())
            return false;
        Operator relu = userOfType(add->getOutput(), OpType::Relu, pos(add));
        if (!relu)
            return false;
        Operator pl = userOfType(relu->getOutput(), OpType::MaxPool, pos(relu));
        if (!pl)
            return false;
        auto pool = as<PoolingObj>(pl);
        const auto [pn, pc, phh, pww, kh, kw] = pool->getNCHWRS();
        const auto [pph, ppw, psh, psw, pdh, pdw] = pool->getPadStrideDilation();
        if (kh != kw || pph != ppw || psh != psw || pdh != 1 || pdw != 1 || pool->getCeilMode() != 0 || ph != pw || sh != sw)
            return false;
        if (!infini_rocm_conv2d_pool_supported(x->getDTypeIndex(), ch, hh, wd, ff, rr, ss, ph, pw, sh, sw, dh, dw, conv->getNumGroups(), 1, kh,
                                               psh, pph))
            return false;
        std::vector<size_t> members{i}, dead;
        std::vector<Read> reads{{x, i}, {w, i}};
        const Tensor root = aliasRoot(other, dead); // the front-end's Reshape(bias, [1, F, 1, 1])
        const Tensor bias = root ? root : other;
        for (size_t dp : dead)
            members.push_back(dp);
        if (!root)
            reads.push_back({other, pos(add)});
        members.push_back(pos(add));
        members.push_back(pos(relu));
        members.push_back(pos(pl));
        std::sort(members.begin(), members.end());
        const size_t slot = pos(pl);
        const Tensor out = pool->getOutput();
        if (overlaps(out, x) || overlaps(out, w) || overlaps(out, bias) || !(out->getDType() == x->getDType()) ||
            !al16((uintptr_t)x->getRawDataPtr<void *>()) || !al16((uintptr_t)out->getRawDataPtr<void *>()) || !readsSurvive(reads, slot, members))
            return false;
        noteLateReads(reads, slot);
        const RocmRuntimeObj *r = R;
        const int n_ = nb, c_ = ch, h_ = hh, w_ = wd, f_ = ff, r_ = rr, s_ = ss, ph_ = ph, pw_ = pw, sh_ = sh, sw_ = sw, dh_ = dh, dw_ = dw;
        const int groups = conv->getNumGroups(), pk = kh, ps = psh, pp = pph;
        emit(slot, members, "conv+bias+relu+maxpool (stem)", true, [=] {
            ConstWeightsScope constWeights(r->handle(), w);
            ROCM_CALL(infini_rocm_conv2d_pool(r->handle(), x->getDTypeIndex(), dataPtr(x), dataPtr(w), dataPtr(bias), dataPtr(out), n_, c_, h_, w_,
                                              f_, r_, s_, ph_, pw_, sh_, sw_, dh_, dw_, groups, 1, pk, ps, pp));
        });
        return true;
    }

    bool planConv(size_t i) {
        auto conv = as<ConvObj>(ops[i]);
        const Tensor x = conv->getInputs(0), w = conv->getInputs(1);
        const auto &od = conv->getOutput()->getDims();
        const int f = od[1];
        // candidate chains, each one op longer than the previous: conv [+ bias] [+ residual] [+ relu]
        struct Cand {
            std::vector<size_t> members;
            std::vector<Read> reads;
            Tensor last, biasSrc, res;
            int act;
            size_t slot;
        };
        std::vector<Cand> cands;
        Cand cur{{i}, {{x, i}, {w, i}}, conv->getOutput(), nullptr, nullptr, 0, i};
        auto add = [&](Cand &c, size_t p) {
            c.members.push_back(p);
            std::sort(c.members.begin(), c.members.end());
            c.slot = std::max(c.slot, p);
        };
        if (Operator u = userOfType(cur.last, OpType::Add, cur.slot)) {
            const Tensor other = otherOf(u, cur.last);
            if (other != cur.last && isChannelBias(other->getDims(), f) && other->getDType() == cur.last->getDType() &&
                u->getOutput()->getDims() == cur.last->getDims()) {
                std::vector<size_t> dead;
                const Tensor root = aliasRoot(other, dead); // the front-end's Reshape(bias, [1, F, 1, 1]): read the weight itself
                cur.biasSrc = root ? root : other;
                for (size_t dp : dead)
                    add(cur, dp);
                if (!root)
                    cur.reads.push_back({other, pos(u)});
                add(cur, pos(u));
                cur.last = u->getOutput();
                cands.push_back(cur);
            }
        }
        // The residual join rides in the conv epilogue where the LDS-staged epilogue serves it (conv_s1.hip: the residual
        // is fetched in the same 128-byte row segments as the stores; even output planes, f16 / bf16). With the earlier
        // direct epilogue (32-byte segments per filter row) the fused form was slower than conv + one ADD_RELU pass, which
        // is still what odd planes (7x7) and fp32 get. INFINI_ROCM_FUSE_RES=0 / =1 forces it off / on for every shape.
        static const int fuseResEnv = std::getenv("INFINI_ROCM_FUSE_RES") ? std::atoi(std::getenv("INFINI_ROCM_FUSE_RES")) : -1;
        // Round 3: pointwise layers that the library runs as a pixel-slot GEMM (csrc/conv.hip: 1 x 1, no padding, C % 64 == 0,
        // >= 128 filters) take the residual on ANY plane, odd ones included (ResNet's 7 x 7 stage: 29 us fused against 24.5 + a
        // 12.7 us ADD_RELU pass).
        bool pixelGemm = false;
        if (od.size() == 4) {
            const auto [nb_, ch_, hh_, wd_, ff_, rr_, ss_] = conv->getNCHWFRS();
            const auto [ph_, pw_, sh_, sw_, dh_, dw_] = conv->getPadStrideDilation();
            pixelGemm = rr_ == 1 && ss_ == 1 && ph_ == 0 && pw_ == 0 && dh_ == 1 && dw_ == 1 && ch_ % 64 == 0 && ff_ >= 128;
        }
        const bool fuseRes = fuseResEnv >= 0 ? fuseResEnv == 1
                                             : (od.size() == 4 && (((long)od[2] * od[3]) % 2 == 0 || pixelGemm) && conv->getNumGroups() == 1 &&
                                                !(x->getDType() == DataType::Float32) && !(x->getDType() == DataType::Double));
        if (fuseRes && cur.biasSrc) { // residual join: the tail of a ResNet bottleneck
            if (Operator u = userOfType(cur.last, OpType::Add, cur.slot)) {
                const Tensor other = otherOf(u, cur.last);
                if (other != cur.last && other->getDims() == cur.last->getDims() && other->getDType() == cur.last->getDType()) {
                    cur.res = other;
                    cur.reads.push_back({other, pos(u)});
                    add(cur, pos(u));
                    cur.last = u->getOutput();
                    cands.push_back(cur);
                }
            }
        }
        if (Operator u = userOfType(cur.last, OpType::Relu, cur.slot)) {
            cur.act = 1;
            add(cur, pos(u));
            cur.last = u->getOutput();
            cands.push_back(cur);
        }
        // longest chain whose output buffer is safe to write while the conv still reads its inputs
        const auto [nb, ch, hh, wd, ff, rr, ss] = conv->getNCHWFRS();
        const auto [ph, pw, sh, sw, dh, dw] = conv->getPadStrideDilation();
        for (auto it = cands.rbegin(); it != cands.rend(); ++it) {
            const Cand &c = *it;
            if (log)
                fprintf(stderr, "[fusion] conv#%zu [%d,%d,%d,%d]: chain %zu (bias %d res %d act %d) out-on-x %d out-on-w %d out-on-res %d\n", i,
                        (int)od[0], (int)od[1], (int)od[2], (int)od[3], c.members.size(), c.biasSrc != nullptr, c.res != nullptr, c.act,
                        (int)overlaps(c.last, x), (int)overlaps(c.last, w), (int)(c.res && overlaps(c.last, c.res)));
            if (!readsSurvive(c.reads, c.slot, c.members))
                continue;
            // the residual is read at exactly the position that is written: it may be the output buffer itself
            const bool resHazard = c.res && overlaps(c.last, c.res) && !samePlace(c.res, c.last);
            // The planner likes to put the chain's output on the conv's own input (dead after the conv in the unfused
            // graph). When that input is small next to the output — the 64 -> 256 expansions of ResNet's first stage: 51 MB
            // in, 205 MB out — the conv reads a copy of it from the workspace instead of giving up the tail: one 2 x 51 MB
            // copy instead of a lone 2 x 205 MB ReLU / bias pass. Unit-stride only (a strided conv keeps its phase planes
            // at the workspace base, a long-K pointwise layer may run as a split-K GEMM with partial planes there); the
            // copy sits behind the conv's own packed-weight area (sized like conv_s1.hip sizes it: [F][roundup32(C R S)] for
            // channel counts that are not multiples of 32).
            const bool onX = overlaps(c.last, x);
            static const bool bridgeOn = envOn("INFINI_ROCM_BRIDGE_X");
            const bool bridgeX = bridgeOn && onX && sh == 1 && sw == 1 && conv->getNumGroups() == 1 && ch < 1024 && ch % 32 == 0 &&
                                 (size_t)x->getBytes() * 3 <= (size_t)c.last->getBytes();
            if (!(c.last->getDType() == x->getDType()))
                continue;
            const bool otherHazard = overlaps(c.last, w) || (c.biasSrc && overlaps(c.last, c.biasSrc)) || resHazard;
            const bool hazard = onX || otherHazard; // (forwarding costs nothing: it is tried before the bridging copy)
            // Buffer forwarding. The chain's planned output buffer is unusable (it sits on something the conv still reads —
            // the memory planner recycles the conv's dead input for the tail's output), but the Conv operator's OWN output
            // buffer was allocated while every operand was live and nobody else needs it once the tail is folded: the fused
            // kernel writes there, and every reader of the chain's final tensor is pointed there (ForwardMap) — provided
            // nothing the planner placed on that (in its eyes free) block is written before the tensor's last reader ran.
            void *fwdTo = nullptr;
            size_t lastUse = c.slot;
            if (hazard) {
                static const bool fwdOn = envOn("INFINI_ROCM_FORWARD");
                const Tensor y = conv->getOutput(), tfin = c.last;
                bool ok = fwdOn && c.members.size() > 1 && !tfin->isOutput() && y->getBytes() == tfin->getBytes() && !overlaps(y, x) &&
                          !overlaps(y, w) && !(c.biasSrc && overlaps(y, c.biasSrc)) && !(c.res && overlaps(y, c.res)) &&
                          !tfin->getTargets().empty();
                for (const auto &u : tfin->getTargets()) {
                    auto it = posOf.find(u.get());
                    // a reader some EARLIER-planned item already owns was checked against the tensor's own buffer (its
                    // in-kernel hazards, the survival of what it reads until its slot): it cannot be re-pointed now
                    ok = ok && it != posOf.end() && it->second > c.slot && !claimed[it->second];
                    if (ok)
                        lastUse = std::max(lastUse, it->second);
                }
                // nothing may land on y's block up to and including the last reader (a reader's own output too: a plain
                // kernel does not know its input moved)
                // a reader's own output too: a plain kernel does not know its input moved), counted from the CONV's position: the
                // planner treats y as free once the bias Add read it, so a non-member operator between the conv and the slot
                // (another branch of an Inception- / SE-style graph) may own y's block, and the fused kernel — which writes y
                // at the slot — would clobber its result (round-3 advisor finding)
                ok = ok && survives(y, i, lastUse + 1, c.members, true);
                if (ok)
                    fwdTo = y->getRawDataPtr<void *>();
                else if (otherHazard || !bridgeX)
                    continue;
            }
            noteLateReads(c.reads, c.slot);
            if (fwdTo) {
                (*fwd)[c.last.get()] = fwdTo;
                lateReads.push_back({i, lastUse, conv->getOutput()}); // y's block stays in use until then
            }
            const RocmRuntimeObj *r = R;
            const Operator op = ops[i];
            const Tensor X = x, Wt = w, Bs = c.biasSrc, Rs = c.res, Out = c.last;
            const bool forwarded = fwdTo != nullptr;
            const int act = c.act, groups = conv->getNumGroups();
            const int variant = tunedVariant(op);
            const int n_ = nb, c_ = ch, h_ = hh, w_ = wd, f_ = ff, r_ = rr, s_ = ss, ph_ = ph, pw_ = pw, sh_ = sh, sw_ = sw, dh_ = dh, dw_ = dw;
            const bool lg = log;
            const size_t idx = i;
            const bool bridge = bridgeX && !forwarded;
            std::string what = std::string("conv") + (c.biasSrc ? "+bias" : "") + (c.res ? "+res" : "") + (c.act ? "+relu" : "") +
                               (bridge ? " (x bridged)" : "") + (forwarded ? " (output forwarded)" : "");
            emit(c.slot, c.members, what, true, [=] {
                const void *xptr = dataPtr(X);
                if (forwarded)
                    ++r->forwardedCount;
                if (bridge) {
                    const size_t kpad = ((size_t)c_ * r_ * s_ + 31) & ~(size_t)31;
                    const size_t wArea = ((std::max((size_t)f_ * c_ * r_ * s_, (size_t)f_ * kpad) * 2 + 4096) + 255) & ~(size_t)255;
                    char *ws = (char *)r->getWorkspace(wArea + X->getBytes());
                    ROCM_CALL(infini_rocm_copy_inside(r->handle(), ws + wArea, xptr, X->getBytes()));
                    xptr = ws + wArea;
                    ++r->bridgedCount;
                    if (lg)
                        fprintf(stderr, "[fusion] conv#%zu: input bridged through the workspace (%zu bytes)\n", idx, (size_t)X->getBytes());
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
                } scope(r->handle(), variant);
                ConstWeightsScope constWeights(r->handle(), Wt); // graph weights: pack once, cache (rocm_runtime.h)
                ROCM_CALL(infini_rocm_conv2d_res(r->handle(), X->getDTypeIndex(), xptr, dataPtr(Wt),
                                                 Bs ? dataPtr(Bs) : nullptr, Rs ? dataPtr(Rs) : nullptr,
                                                 dataPtr(Out), n_, c_, h_, w_, f_, r_, s_, ph_, pw_, sh_, sw_, dh_, dw_, groups,
                                                 act));
            },
                 forwarded ? std::vector<size_t>{i} : std::vector<size_t>{}); // forwarded: the Conv operator's own buffer is the one written
            return true;
        }
        return false;
    }

    // ============================================================================================================
    // element-wise / normalisation pairs
    // ============================================================================================================
    // Silu(a) -> Mul(., b): the gate of a gated MLP as one pass (infini_rocm_silu_mul; bit-identical: the Silu value is rounded
    // as its own kernel would have stored it). Operators between the two that a grouped launch already ran (the up projection)
    // are simply not in the way; b may be a result parked in the workspace by that launch.
    bool planSiluMul(size_t i) {
        static const bool on = envOn("INFINI_ROCM_FUSE_SWIGLU");
        const Operator op = ops[i];
        Operator mul = on ? userOfType(op->getOutput(), OpType::Mul, i) : nullptr;
        if (!mul)
            return false;
        const size_t m = pos(mul);
        const Tensor a = op->getInputs(0), sOut = op->getOutput(), out = mul->getOutput();
        if (!usesOnce(mul, sOut))
            return false;
        const Tensor b = otherOf(mul, sOut);
        const int dt = a->getDTypeIndex();
        auto safe = [&](const Tensor &u) { return !overlaps(out, u) || samePlace(u, out); }; // element-wise: exactly in place is fine
        auto pf = parked.find(m);
        const bool bParked = pf != parked.end() && pf->second.tensor == b.get();
        if (pf != parked.end() && !bParked)
            return false;
        if (!(b != sOut && (isHalf(dt) || dt == INFINI_DT_F32) && a->getDims() == out->getDims() && b->getDims() == out->getDims() &&
              b->getDType() == a->getDType() && out->getDType() == a->getDType() && safe(a) && (bParked || safe(b)) &&
              al16(addrOf(a) | addrOf(out) | (bParked ? 0 : addrOf(b)))))
            return false;
        std::vector<size_t> members{i, m};
        std::vector<Read> reads{{a, i}};
        if (!readsSurvive(reads, m, members))
            return false;
        noteLateReads(reads, m);
        const RocmRuntimeObj *r = R;
        const size_t pbytes = bParked ? pf->second.bytes : 0;
        if (bParked)
            parked.erase(pf);
        emit(m, members, bParked ? "silu_mul(parked b)" : "silu_mul", true, [r, a, b, out, dt, bParked, pbytes] {
            const void *bp = bParked ? r->getWorkspace(pbytes) : dataPtr(b);
            ROCM_CALL(infini_rocm_silu_mul(r->handle(), dt, dataPtr(a), bp, dataPtr(out), (int64_t)out->size()));
        });
        return true;
    }

    // Relu -> MaxPool: max and relu commute (bit-identical); the stem of every ResNet
    bool planReluPool(size_t i) {
        const Operator op = ops[i];
        Operator pl = userOfType(op->getOutput(), OpType::MaxPool, i);
        if (!pl)
            return false;
        auto pool = as<PoolingObj>(pl);
        const Tensor x = op->getInputs(0), out = pool->getOutput();
        std::vector<size_t> members{i, pos(pl)};
        if (overlaps(out, x) || !readsSurvive({{x, i}}, pos(pl), members))
            return false;
        noteLateReads({{x, i}}, pos(pl));
        const auto [nb, c, h, w, kh, kw] = pool->getNCHWRS();
        const auto [ph, pw, sh, sw, dh, dw] = pool->getPadStrideDilation();
        const int ceil = pool->getCeilMode();
        const RocmRuntimeObj *r = R;
        const int n_ = nb, c_ = c, h_ = h, w_ = w, kh_ = kh, kw_ = kw, ph_ = ph, pw_ = pw, sh_ = sh, sw_ = sw, dh_ = dh, dw_ = dw;
        emit(pos(pl), members, "relu+maxpool", true, [=] {
            ROCM_CALL(infini_rocm_pool2d_relu(r->handle(), 0, x->getDTypeIndex(), dataPtr(x), dataPtr(out), n_,
                                              c_, h_, w_, kh_, kw_, dh_, dw_, ph_, pw_, sh_, sw_, ceil, 1));
        });
        return true;
    }

    // Add(a, b) (same extents) -> LayerNormalization over the last axis / RMSNorm / the nine-operator LayerNorm: one pass.
    // Also Add(a, row bias) -> Add(., b) -> Norm: a linear layer's bias that could not ride in the GEMM epilogue (the
    // memory planner likes to put the bias Add's output on the MatMul's dead A operand) joins the residual add and the
    // normalisation instead (infini_rocm_bias_add_norm).
    bool planAddNorm(size_t i) {
        const Operator op = ops[i];
        Tensor a = op->getInputs(0), b = op->getInputs(1), t = op->getOutput(), pre = nullptr;
        const Shape td = t->getDims();
        if (!(a->getDType() == b->getDType()) || t->isOutput() || td.empty())
            return false;
        std::vector<size_t> members{i};
        std::vector<Read> reads;
        size_t sumAt = i; // position of the Add whose result is normalised
        if (!(a->getDims() == td && b->getDims() == td)) {
            static const bool on = envOn("INFINI_ROCM_FUSE_BIAS_NORM");
            const bool aRow = isRowVector(a->getDims(), td.back()), bRow = isRowVector(b->getDims(), td.back());
            const Tensor full = aRow ? b : a, row = aRow ? a : b;
            if (!on || aRow == bRow || full->getDims() != td || (int)td.size() < 2)
                return false;
            Operator add2 = userOfType(t, OpType::Add, i);
            if (!add2)
                return false;
            const Tensor res = otherOf(add2, t);
            if (res == t || res->getDims() != td || !(res->getDType() == t->getDType()) || add2->getOutput()->getDims() != td ||
                add2->getOutput()->isOutput())
                return false;
            std::vector<size_t> dead;
            const Tensor root = aliasRoot(row, dead);
            pre = root ? root : row;
            for (size_t dp : dead)
                if (dp > i)
                    members.push_back(dp);
            if (!root)
                reads.push_back({row, i});
            members.push_back(pos(add2));
            reads.push_back({full, i});
            reads.push_back({res, pos(add2)});
            a = full;
            b = res;
            t = add2->getOutput();
            sumAt = pos(add2);
        } else {
            reads = {{a, i}, {b, i}};
        }
        Tensor scale, bias, out;
        float eps = 1e-5f; // RMSNorm: hard-coded in the reference (rms_norm.cu:46)
        bool rms = false;
        size_t slot = sumAt;
        std::string what = pre ? "bias+" : "";
        Operator nrm = onlyUser(t);
        if (nrm && pos(nrm) > sumAt && (nrm->getOpType() == OpType::LayerNormalization || nrm->getOpType() == OpType::RMSNorm) &&
            nrm->getInputs(0) == t) {
            rms = nrm->getOpType() == OpType::RMSNorm;
            scale = nrm->getInputs(1);
            if (!rms) {
                auto ln = as<LayerNormObj>(nrm);
                if (ln->getAxis() != (int)td.size() - 1)
                    return false;
                eps = ln->getEps();
                if (ln->numInputs() == 3)
                    bias = ln->getInputs(2);
            }
            out = nrm->getOutput();
            members.push_back(pos(nrm));
            slot = pos(nrm);
            what += rms ? "add+rmsnorm" : "add+layernorm";
        } else {
            NormMatch m;
            if (!matchLayerNormDecomposed(t, m) || t->getTargets().size() != 2 || m.first < sumAt)
                return false;
            scale = m.gamma;
            bias = m.beta;
            eps = (float)m.eps;
            out = m.out;
            members.insert(members.end(), m.members.begin(), m.members.end());
            slot = m.last;
            what += "add+layernorm(decomposed)";
        }
        std::sort(members.begin(), members.end());
        auto hazard = [&](const Tensor &u) { return overlaps(out, u) && !samePlace(u, out); }; // row-wise in place is safe
        if (hazard(a) || hazard(b) || overlaps(out, scale) || (bias && overlaps(out, bias)) || (pre && overlaps(out, pre)) ||
            !(out->getDType() == t->getDType()))
            return false;
        if (!readsSurvive(reads, slot, members))
            return false;
        noteLateReads(reads, slot);
        const int64_t nn = td.back(), outer = (int64_t)t->size() / nn;
        const RocmRuntimeObj *r = R;
        const int dt = t->getDTypeIndex();
        emit(slot, members, what, true, [r, a, b, pre, scale, bias, out, outer, nn, eps, rms, dt] {
            ROCM_CALL(infini_rocm_bias_add_norm(r->handle(), dt, rms ? 1 : 0, dataPtr(a),
                                                pre ? dataPtr(pre) : nullptr, dataPtr(b),
                                                dataPtr(scale), bias ? dataPtr(bias) : nullptr,
                                                dataPtr(out), outer, nn, (int64_t)scale->size(),
                                                bias ? (int64_t)bias->size() : 0, eps));
        });
        return true;
    }

    // Add(x, per-channel bias) -> Add(., identity) [-> Relu]: the bottleneck tail when the bias could not ride in the conv
    bool planBiasResidual(size_t i) {
        const Operator op = ops[i];
        const Tensor t = op->getOutput();
        const Tensor a0 = op->getInputs(0), a1 = op->getInputs(1);
        const auto &td = t->getDims();
        Operator add2 = userOfType(t, OpType::Add, i);
        if (!add2 || td.size() < 2)
            return false;
        Tensor xin = nullptr, biasT = nullptr;
        if (isChannelBiasOf(a1->getDims(), td) && a0->getDims() == td) { xin = a0; biasT = a1; }
        else if (isChannelBiasOf(a0->getDims(), td) && a1->getDims() == td) { xin = a1; biasT = a0; }
        if (!xin)
            return false;
        std::vector<size_t> members{i}, dead;
        const Tensor root = aliasRoot(biasT, dead);
        const Tensor biasSrc = root ? root : biasT;
        for (size_t dp : dead)
            if (dp > i) // an alias operator in front of the chain's head already ran (or will): leave it alone
                members.push_back(dp);
        const Tensor res = otherOf(add2, t);
        Tensor out = add2->getOutput();
        members.push_back(pos(add2));
        size_t slot = pos(add2);
        int relu = 0;
        if (!(res != t && res->getDims() == td && res->getDType() == t->getDType() && biasT->getDType() == t->getDType()))
            return false;
        if (Operator rl = userOfType(out, OpType::Relu, slot)) {
            out = rl->getOutput();
            relu = 1;
            members.push_back(pos(rl));
            slot = pos(rl);
        }
        auto hazard = [&](const Tensor &u) { return overlaps(out, u) && !samePlace(u, out); };
        if (hazard(xin) || hazard(res) || overlaps(out, biasSrc))
            return false;
        std::sort(members.begin(), members.end());
        std::vector<Read> reads{{xin, i}, {res, pos(add2)}};
        if (!root)
            reads.push_back({biasT, i});
        if (!readsSurvive(reads, slot, members))
            return false;
        noteLateReads(reads, slot);
        int64_t inner = 1;
        for (size_t d = 2; d < td.size(); ++d)
            inner *= td[d];
        const RocmRuntimeObj *r = R;
        const int dt = t->getDTypeIndex();
        const int64_t d0 = td[0], d1 = td[1];
        emit(slot, members, relu ? "bias+res+relu" : "bias+res", true, [r, xin, biasSrc, res, out, d0, d1, inner, relu, dt] {
            ROCM_CALL(infini_rocm_bias_residual(r->handle(), dt, dataPtr(xin), dataPtr(biasSrc),
                                                dataPtr(res), dataPtr(out), d0, d1, inner, relu));
        });
        return true;
    }

    bool planAddRelu(size_t i) {
        const Operator op = ops[i];
        Operator rl = userOfType(op->getOutput(), OpType::Relu, i);
        if (!rl)
            return false;
        const Tensor a = op->getInputs(0), b = op->getInputs(1), out = rl->getOutput();
        const Shape od = op->getOutput()->getDims();
        // element-wise with identical extents may run exactly in place (the planner likes to give the Relu output the
        // storage of a dead Add input); any other overlap is a hazard
        auto hazard = [&](const Tensor &t) { return overlaps(out, t) && !(addrOf(t) == addrOf(out) && t->getDims() == od); };
        if (!(a->getDType() == b->getDType()) || hazard(a) || hazard(b))
            return false;
        std::vector<size_t> members{i, pos(rl)};
        std::vector<Read> reads{{a, i}, {b, i}};
        if (!readsSurvive(reads, pos(rl), members))
            return false;
        noteLateReads(reads, pos(rl));
        const RocmRuntimeObj *r = R;
        emit(pos(rl), members, "add+relu", true, [r, a, b, out, od] {
            const auto shape = std::vector<int64_t>(od.begin(), od.end());
            const auto sa = strides64(a->getDims(), od), sb = strides64(b->getDims(), od);
            ROCM_CALL(infini_rocm_binary(r->handle(), INFINI_BIN_ADD_RELU, a->getDTypeIndex(), dataPtr(a),
                                         dataPtr(b), dataPtr(out), (int)shape.size(), shape.data(), sa.data(),
                                         sb.data()));
        });
        return true;
    }

    // RoPE -> Reshape([B, S, H, D]) -> Transpose(0, 2, 1, 3): the rotary embedding of a decoder's q / k followed by their head split
    // (rope.cu, reshape.cc, transpose.cc: three launches, two extra passes) as one pass with a head-split store
    // (infini_rocm_rope_headsplit). Head dim 128 / theta 1e4 as the reference hard-codes them (rope.cc:25). The input may be a
    // grouped MatMul's result parked in the workspace.
    bool planRopeHeadSplit(size_t i, const ParkedFeed *pf) {
        static const bool on = envOn("INFINI_ROCM_FUSE_ROPE_SPLIT");
        if (!on)
            return false;
        const Operator op = ops[i];
        const Tensor posT = op->getInputs(0), x = op->getInputs(1), y = op->getOutput();
        Operator rs = userOfType(y, OpType::Reshape, i);
        Operator tr = rs ? userOfType(rs->getOutput(), OpType::Transpose, pos(rs)) : nullptr;
        if (!tr)
            return false;
        const Tensor r4 = rs->getOutput(), out = tr->getOutput();
        const auto &xd = x->getDims(), &rd = r4->getDims();
        if (xd.size() != 3 || rd.size() != 4 || !permIs(as<TransposeObj>(tr)->getPermute(), 0, 2, 1, 3) || rd[0] != xd[0] || rd[1] != xd[1] ||
            rd[3] != 128 || (long)rd[2] * rd[3] != xd[2] || posT->getDims().size() != 2 || posT->getDims()[1] != xd[1] ||
            !(out->getDType() == x->getDType()) || out->getBytes() != x->getBytes())
            return false;
        if (pf && pf->tensor != x.get())
            return false;
        if ((!pf && overlaps(out, x)) || overlaps(out, posT))
            return false;
        std::vector<size_t> members{i, pos(rs), pos(tr)};
        const size_t slot = pos(tr);
        std::vector<Read> reads{{posT, i}};
        if (!pf)
            reads.push_back({x, i});
        else if (slot != i + 2)
            return false; // the parked copy lives in the workspace: nothing may run in between
        if (!readsSurvive(reads, slot, members))
            return false;
        noteLateReads(reads, slot);
        const RocmRuntimeObj *r = R;
        const bool isParked = pf != nullptr;
        const size_t pbytes = pf ? pf->bytes : 0;
        const int64_t rows = (int64_t)xd[0] * xd[1], width = xd[2], seq = xd[1];
        if (pf)
            parked.erase(i);
        emit(slot, members, isParked ? "rope+headsplit(parked x)" : "rope+headsplit", true, [r, x, posT, out, rows, width, seq, isParked, pbytes] {
            const void *xp = isParked ? r->getWorkspace(pbytes) : dataPtr(x);
            ROCM_CALL(infini_rocm_rope_headsplit(r->handle(), x->getDTypeIndex(), posT->getDTypeIndex(), dataPtr(posT), xp,
                                                 dataPtr(out), rows, width, 128, 10000.0f, seq));
        });
        return true;
    }

    // producer -> Reshape | Flatten | Identity | Squeeze | Unsqueeze: the reference runs these as a device memcpy
    // (CopyCuda, reshape.cc:4-21); here the producer writes straight into the copy's output buffer. The producer runs through
    // its normal kernel (and perf record) with its output tensor redirected, so nothing about its numerics changes. A copy of
    // a weight / graph input whose every reader was folded is handled by the readers (aliasRoot), not here.
    bool planIntoCopy(size_t i) {
        static const bool on = envOn("INFINI_ROCM_FUSE_RESHAPE");
        const Operator op = ops[i];
        const auto type = op->getOpType();
        // producers whose kernels only WRITE their output (no in-place state, no multi-output, no collectives)
        if (!on || !(type == OpType::MatMul || type == OpType::Transpose || type == OpType::Add || type == OpType::Sub || type == OpType::Mul ||
                     type == OpType::Div || type == OpType::Relu || type == OpType::Gelu || type == OpType::Silu || type == OpType::Sigmoid ||
                     type == OpType::Tanh || type == OpType::Softmax || type == OpType::LayerNormalization || type == OpType::RMSNorm ||
                     type == OpType::Gather || type == OpType::RoPE))
            return false;
        if (op->numOutputs() != 1)
            return false;
        const Tensor mid = op->getOutput();
        Operator next = onlyUser(mid);
        if (!next || pos(next) <= i || !isCopyLike(next->getOpType()) || next->numOutputs() != 1 || next->getInputs(0) != mid)
            return false;
        const Tensor out = next->getOutput();
        if (mid->getBytes() != out->getBytes() || !(mid->getDType() == out->getDType()))
            return false;
        std::vector<Read> reads;
        for (const auto &in : op->getInputs()) {
            if (overlaps(out, in)) // the producer would overwrite what it is still reading
                return false;
            reads.push_back({in, i});
        }
        std::vector<size_t> members{i, pos(next)};
        if (!readsSurvive(reads, pos(next), members))
            return false;
        noteLateReads(reads, pos(next));
        const RocmRuntimeObj *r = R;
        emit(pos(next), members, ">reshape", true, [r, op, mid, out] {
            OverrideScope s;
            s.redirect(mid.get(), dataPtr(out));
            r->launchOne(op);
        });
        return true;
    }
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
