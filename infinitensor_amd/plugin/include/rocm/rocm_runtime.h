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
#include <list>
#include <memory>
#include <mutex>
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

    void initComm(const string &name, int worldSize, int rank) final;
    CommunicatorObj &getCommunicator() const final;

    // Output redirection for the "producer -> Reshape" fusion (rocm_fusion.cc): while set, the kernels of
    // rocm_kernels.cc resolve `redirectTensor` to `redirectPtr` instead of the tensor's own buffer, so the producer
    // writes straight into the Reshape's output and the copy (reference: CopyCuda, reshape.cc:4-13) is not launched.
    static thread_local const TensorObj *redirectTensor;
    static thread_local void *redirectPtr;
    // with a redirected MatMul: store the result head-split ([m / seq][n / headDim][seq][headDim]) — the
    // MatMul -> Reshape -> Transpose(0, 2, 1, 3) fusion (infini_rocm_matmul_headsplit); 0 = plain store
    static thread_local int redirectSeq, redirectHeadDim;
    // with a redirected MatMul: activation applied in the GEMM epilogue (the MatMul -> Gelu fusion passes 5); 0 = none
    static thread_local int redirectAct;

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

    void launchAll(const Graph &graph, bool validate) const;
    // launches ops[i .. i+k) as one kernel when a fusion rule applies; returns k (0 = no rule)
    size_t tryLaunchFused(const OpVec &ops, size_t i) const;
    size_t tryLaunchGroupedMatmul(const OpVec &ops, size_t i) const;
    size_t tryLaunchRopeHeadSplit(const OpVec &ops, size_t i) const;
    void launchWithInputRedirect(const Operator &op, const TensorObj *t, void *ptr) const;
    size_t tryLaunchFusedRules(const OpVec &ops, size_t i) const;
    size_t tryLaunchIntoReshape(const OpVec &ops, size_t i) const;
    size_t tryLaunchHeadSplit(const OpVec &ops, size_t i) const;
    size_t tryLaunchMatmulGelu(const OpVec &ops, size_t i) const;
    void launchOne(const Operator &op) const; // one operator through KernelRegistry (+ its perf record, if tuned)
    size_t tryLaunchFusedAttention(const OpVec &ops, size_t i) const;
    int tunedVariant(const Operator &op) const; // kernel variant chosen by tune() for this operator's workload, or -1
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
    mutable std::vector<char> launchedAhead; // per launchAll: operators a grouped launch already ran out of order
    // per launchAll: operator index -> (tensor, pointer): this operator reads `tensor` from `pointer` (a result a grouped launch
    // parked in the workspace because the tensor's own buffer was still in use when the group ran); launched without fusion
    struct ParkedFeed {
        const TensorObj *tensor;
        void *ptr;
    };
    mutable std::map<size_t, ParkedFeed> parkedFeeds;
    mutable size_t parkedCount = 0;
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
