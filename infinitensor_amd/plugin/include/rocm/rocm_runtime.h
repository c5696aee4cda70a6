// Device::ROCM runtime for the InfiniTensor graph executor — the MI355X peer of CudaRuntimeObj
// (reference: include/cuda/cuda_runtime.h:12-140, src/cuda/cuda_runtime.cc). It is a thin C++ shell
// over the C ABI in include/infini_rocm.h: the handle owns the device, one non-blocking HIP stream,
// the scratch workspace and (optionally) one RCCL communicator; this class adds what only the
// graph layer knows — walking the operator list through KernelRegistry, and the hipGraph
// capture/replay cache keyed by the graph's capture state (same invalidation rules as the reference's
// CUDA-graph cache, cuda_runtime.cc:210-426).
#pragma once
#include <map>
#include <vector>
#include "core/communicator.h"
#include "core/runtime.h"
#include "core/tensor.h"
#include "infini_rocm.h"
#include <functional>
#include <list>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>

namespace infini {

// reference: NcclCommunicatorObj (include/cuda/nccl_communicator.h:22-68). The RCCL communicator
// itself lives behind the C ABI (one per runtime handle); this object only carries world/rank.
class RcclCommunicatorObj final : public CommunicatorObj {
  public:
    RcclCommunicatorObj(int worldSize, int rank) : CommunicatorObj(worldSize, rank) {}
    string toString() const final { return "RCCL communicator"; }
};

class RocmRuntimeObj : public RuntimeObj {
  public:
    explicit RocmRuntimeObj(int deviceId = 0, size_t hipGraphCacheCapacity = 16);
    ~RocmRuntimeObj() override;
    string toString() const override;

    void run(const Graph &graph, bool tune = false, bool profiling = false) const override;
    void runWithoutSync(const Graph &graph) const;
    void sync() const;
    // capture-once / replay (reference: runWithCudaGraph)
    void runWithHipGraph(const Graph &graph);
    void clearHipGraphCache();
    size_t getHipGraphCacheSize() const;
    size_t getHipGraphCaptureCount() const;
    void invalidateGraphCaptureCache(uint64_t graphId) noexcept override;

    void *alloc(size_t size) override;
    void dealloc(void *ptr) override;
    void copyBlobFromCPU(void *dst, const void *src, size_t bytes) const override;
    void copyBlobToCPU(void *dst, const void *src, size_t bytes) const override;
    void copyBlobInsideRuntime(void *dst, const void *src, size_t bytes) const override;

    // scratch valid inside one Kernel::compute (reference: getWorkspace, cuda_runtime.h:85-88)
    void *getWorkspace(size_t size) const;
    infiniRocmRuntime_t handle() const { return rt; }

    // launch-time fusion of element-wise tails into the producing kernel (src/rocm_fusion.cc); default on,
    // INFINI_ROCM_FUSION=0 in the environment turns it off at construction
    void setFusion(bool on);
    bool getFusion() const { return fusion; }
    size_t getFusedLaunchCount() const { return fusedCount; } // fused kernels launched so far (tests)
    size_t getBridgedInputCount() const { return bridgedCount; } // fused convs that read a workspace copy of their input
    size_t getParkedMemberCount() const { return parkedCount; }   // grouped MatMul members whose result was parked in the workspace

    // the autotune cache: h.tune() fills the process-wide PerfEngine (MatMul / Conv pick among their kernel variants,
    // rocm/rocm_perf.h); these persist it as JSON and bring it back (reference: PerfEngine::savePerfEngineData /
    // loadPerfEngineData, src/core/perf_engine.cc:7-21 — not exported to Python there)
    static void savePerfData(const string &path);
    static void loadPerfData(const string &path);
    static void clearPerfData();
    static size_t perfDataSize();
    static size_t perfEpoch; // bumped whenever the autotune records change (tune(), load_perf, clear_perf): plans embed the picks

    void initComm(const string &name, int worldSize, int rank) final;
    CommunicatorObj &getCommunicator() const final;

    // Per-launch overrides a planned (fused) launch hands to the kernels of rocm_kernels.cc (thread-local, set by an RAII
    // scope around ONE launchOne / C-ABI call, never alive across launches):
    //  * tensor[i] -> ptr[i]: P(t) resolves these tensors to the given pointers instead of their own buffers — a producer
    //    writing straight into a Reshape's output (reference: CopyCuda, reshape.cc:4-13, not launched), or an operator
    //    reading a result that a previous launch left in the workspace;
    //  * for the MatMul `matmul`: a folded Add(bias) (`biasPtr`: a row vector of n elements), an activation in the GEMM
    //    epilogue (`act`, 5 = Gelu) and a head-split store ([m / seq][n / headDim][seq][headDim]: the
    //    MatMul -> Reshape -> Transpose(0, 2, 1, 3) chain, infini_rocm_matmul_headsplit); 0 = plain.
    struct LaunchOverrides {
        const TensorObj *tensor[2] = {nullptr, nullptr};
        void *ptr[2] = {nullptr, nullptr};
        const OperatorObj *matmul = nullptr;
        const void *biasPtr = nullptr;
        int act = 0, seq = 0, headDim = 0;
    };
    static thread_local LaunchOverrides overrides;

    // One entry of a launch plan (src/rocm_fusion.cc): the operators `members` (positions in the graph's operator order)
    // run as ONE launch sequence `run` at position `slot`.
    struct PlanItem {
        size_t slot = 0;
        std::vector<size_t> members;
        std::string what;           // "conv+bias+relu", "matmul+bias+headsplit x3", ... ("" = the operator's own kernel)
        std::function<void()> run;  // empty: nothing to launch (the operators were absorbed by another item)
        bool fused = false;
    };
    // Buffer forwarding: a tensor whose value a planned launch left in ANOTHER buffer than the tensor's own (the fused conv
    // chain whose planned output buffer overlaps the conv's input writes into the Conv operator's own, otherwise unused,
    // output buffer instead); every reader of the tensor — planned items and plain kernels (P() in rocm_kernels.cc) —
    // resolves it through this map while the plan is made and while it runs.
    using ForwardMap = std::unordered_map<const TensorObj *, void *>;
    static thread_local const ForwardMap *forwards;
    struct LaunchPlan {
        std::vector<PlanItem> items;
        std::shared_ptr<ForwardMap> forwarded;
    };
    size_t getForwardedCount() const { return forwardedCount; } // fused chains that wrote into a forwarded buffer (tests)
    // What launchAll would do for `graph` (any runtime: on a non-ROCM runtime nothing can be launched, the plan is only
    // described) — one line per item: "<slot> <what> [members]". Tests and tools/plan_dump.py read it.
    static std::vector<std::string> describeFusionPlan(const Graph &graph);

  private:
    struct TensorState {
        const void *tensor;
        int dtype;
        vector<int> shape;
        uint64_t storageId;
        size_t storageOffset;
        const void *address;
        bool operator==(const TensorState &o) const {
            return tensor == o.tensor && dtype == o.dtype && shape == o.shape && storageId == o.storageId &&
                   storageOffset == o.storageOffset && address == o.address;
        }
    };
    struct GraphState {
        uint64_t graphId;
        size_t topologyEpoch;
        vector<TensorState> tensors;
        bool operator==(const GraphState &o) const {
            return graphId == o.graphId && topologyEpoch == o.topologyEpoch && tensors == o.tensors;
        }
    };
    struct CacheEntry {
        WRef<GraphObj> owner;
        GraphState state;
        size_t generation;
        infiniRocmGraph_t graph = nullptr;
        ~CacheEntry();
    };
    using Cache = std::list<std::unique_ptr<CacheEntry>>;

    friend class FusionPlanner; // src/rocm_fusion.cc
    void launchAll(const Graph &graph, bool validate) const;
    // the launch plan of a graph: which operators run fused, where (src/rocm_fusion.cc); fusion off = one item per operator
    LaunchPlan buildPlan(const Graph &graph) const;
    // the plan of `graph`, re-made only when something it was decided on changed: the graph's capture generation (topology,
    // shapes, storage — the same notion the hipGraph fast path uses), the fusion switch, the autotune records, a constant the
    // planner read, the communicator
    std::shared_ptr<const LaunchPlan> planOf(const Graph &graph) const;
    void dropPlans() const;
    void executePlan(const LaunchPlan &plan, const OpVec &ops) const;
    void launchOne(const Operator &op) const; // one operator through KernelRegistry (+ its perf record, if tuned)
    // a host write went over [ptr, ptr + bytes): scalar constants the planner read from there are stale (returns whether
    // any were — captured graphs embed plans that were decided on those values)
    bool forgetScalars(const void *ptr, size_t bytes) const;
    void tuneImpl(const Graph &graph, bool profiling) const;
    GraphState stateOf(const Graph &graph) const;
    void replay(CacheEntry &entry);
    // a host copy dropped a packed weight image that captured graphs may read: forget every capture
    void dropCapturesIfWeightsChanged(uint64_t epochBefore) const;

    infiniRocmRuntime_t rt = nullptr;
    std::unique_ptr<CommunicatorObj> comm;
    size_t cacheCapacity;
    size_t captureCount = 0;
    bool fusion = true;
    mutable size_t fusedCount = 0;
    mutable size_t bridgedCount = 0;
    mutable size_t parkedCount = 0;
    mutable size_t forwardedCount = 0;
    // values of one-element constant tensors (weights no operator writes) the planner looked at — Pow's exponent, the
    // sqrt(2) / 0.5 / 1 of a decomposed Gelu, LayerNorm's epsilon — keyed by device address; dropped by host writes
    mutable std::map<const void *, std::pair<size_t, double>> scalarCache;
    struct PlanEntry {
        WRef<GraphObj> owner;
        uint64_t graphId = 0;
        size_t generation = 0, perfEpoch = 0;
        bool fusion = true;
        std::shared_ptr<const LaunchPlan> plan;
    };
    mutable std::map<const GraphObj *, PlanEntry> plans;
    Cache cache; // most recently used first
    mutable std::recursive_mutex executionMutex;
    mutable std::recursive_mutex cacheMutex;
};

// While alive: the conv kernels treat the weight tensor as constant data and keep its re-packed image in the runtime's
// packed-weight cache (infini_rocm_conv2d_set_const_weights). Only for weights no operator of the graph writes (graph
// weights / inputs: getSource() == nullptr) — whatever the host copies over them goes through copyBlobFromCPU /
// copyBlobInsideRuntime, which drop the stale image (and, in RocmRuntimeObj, the captured graphs that read it).
struct ConstWeightsScope {
    infiniRocmRuntime_t rt;
    bool on;
    ConstWeightsScope(infiniRocmRuntime_t rt, const Tensor &w) : rt(rt), on(w->getSource() == nullptr) {
        if (on)
            (void)infini_rocm_conv2d_set_const_weights(rt, 1);
    }
    ~ConstWeightsScope() {
        if (on)
            (void)infini_rocm_conv2d_set_const_weights(rt, 0);
    }
};

// Throw infini::Exception with the C ABI's message when a call fails
// (reference: checkCudaError, include/cuda/cuda_common.h:10-14).
void rocmCheck(int status, const char *what);
#define ROCM_CALL(expr) ::infini::rocmCheck((expr), #expr)

} // namespace infini
