#include "rocm/rocm_runtime.h"
#include <cstdlib>
#include "core/graph.h"
#include "core/kernel.h"
#include "core/perf_engine.h"
#include <algorithm>

namespace infini {

void rocmCheck(int status, const char *what) {
    if (status != INFINI_ROCM_OK) {
        const std::string msg = std::string(what) + " failed (status " + std::to_string(status) + "): " +
                                infini_rocm_last_error();
        // `<< msg`: without BACKWARD_TRACE the reference's Exception leaves what() empty (utils/exception.cc)
        throw(::infini::Exception(msg) << msg);
    }
}

RocmRuntimeObj::RocmRuntimeObj(int deviceId, size_t hipGraphCacheCapacity)
    : RuntimeObj(Device::ROCM, deviceId), cacheCapacity(hipGraphCacheCapacity) {
    IT_ASSERT(hipGraphCacheCapacity > 0, "hipGraph cache capacity must be greater than zero"); // cuda_runtime.cc:78
    ROCM_CALL(infini_rocm_runtime_create(deviceId, &rt));
    if (const char *e = std::getenv("INFINI_ROCM_FUSION"))
        fusion = std::string(e) != "0";
}

void RocmRuntimeObj::setFusion(bool on) {
    std::lock_guard<std::recursive_mutex> lock(executionMutex);
    if (on != fusion) {
        fusion = on;
        clearHipGraphCache(); // captured launch sequences embed the fusion decisions
    }
}

RocmRuntimeObj::~RocmRuntimeObj() {
    cache.clear();
    plans.clear();
    if (rt)
        infini_rocm_runtime_destroy(rt);
}

string RocmRuntimeObj::toString() const { return "ROCM Runtime"; }

void *RocmRuntimeObj::alloc(size_t size) {
    void *p = nullptr;
    ROCM_CALL(infini_rocm_alloc(rt, size, &p));
    return p;
}
void RocmRuntimeObj::dealloc(void *ptr) { ROCM_CALL(infini_rocm_dealloc(rt, ptr)); }
static uint64_t weightEpoch(infiniRocmRuntime_t rt) {
    uint64_t e = 0;
    (void)infini_rocm_weight_cache_info(rt, nullptr, nullptr, &e);
    return e;
}
void RocmRuntimeObj::dropCapturesIfWeightsChanged(uint64_t epochBefore) const {
    if (weightEpoch(rt) == epochBefore)
        return;
    // the copy went over a conv weight whose packed image was cached: captured graphs still address the old image
    auto *self = const_cast<RocmRuntimeObj *>(this);
    std::lock_guard<std::recursive_mutex> cacheLock(cacheMutex);
    self->cache.clear();
    // no graph exec of this runtime is alive any more: the dropped image (and outgrown scratch blocks) can be freed now —
    // a workload that re-uploads a weight every run would otherwise retire one image per run until the device is full
    (void)infini_rocm_workspace_trim(rt);
}
bool RocmRuntimeObj::forgetScalars(const void *ptr, size_t bytes) const {
    if (scalarCache.empty())
        return false;
    bool any = false;
    const char *lo = (const char *)ptr, *hi = lo + bytes;
    for (auto it = scalarCache.begin(); it != scalarCache.end();) {
        const char *p = (const char *)it->first;
        if (p < hi && lo < p + it->second.first) {
            it = scalarCache.erase(it);
            any = true;
        } else {
            ++it;
        }
    }
    return any;
}
void RocmRuntimeObj::copyBlobFromCPU(void *dst, const void *src, size_t bytes) const {
    const uint64_t e0 = weightEpoch(rt);
    ROCM_CALL(infini_rocm_copy_from_cpu(rt, dst, src, bytes));
    dropCapturesIfWeightsChanged(e0);
    if (forgetScalars(dst, bytes)) { // a constant a launch plan was decided on (Pow's exponent, a Gelu's sqrt 2): re-plan
        std::lock_guard<std::recursive_mutex> cacheLock(cacheMutex);
        const_cast<RocmRuntimeObj *>(this)->cache.clear();
        dropPlans();
    }
}
void RocmRuntimeObj::copyBlobToCPU(void *dst, const void *src, size_t bytes) const {
    ROCM_CALL(infini_rocm_copy_to_cpu(rt, dst, src, bytes));
}
void RocmRuntimeObj::copyBlobInsideRuntime(void *dst, const void *src, size_t bytes) const {
    const uint64_t e0 = weightEpoch(rt);
    ROCM_CALL(infini_rocm_copy_inside(rt, dst, src, bytes));
    dropCapturesIfWeightsChanged(e0);
    if (forgetScalars(dst, bytes)) {
        std::lock_guard<std::recursive_mutex> cacheLock(cacheMutex);
        const_cast<RocmRuntimeObj *>(this)->cache.clear();
        dropPlans();
    }
    ROCM_CALL(infini_rocm_runtime_sync(rt)); // reference semantics: cudaMemcpy D2D is synchronous
}
void *RocmRuntimeObj::getWorkspace(size_t size) const {
    void *p = nullptr;
    ROCM_CALL(infini_rocm_workspace(rt, size, &p));
    return p;
}
void RocmRuntimeObj::sync() const { ROCM_CALL(infini_rocm_runtime_sync(rt)); }

void RocmRuntimeObj::initComm(const string &name, int worldSize, int rank) {
    std::lock_guard<std::recursive_mutex> lock(executionMutex);
    IT_ASSERT(worldSize > 0);
    IT_ASSERT(rank >= 0 && rank < worldSize);
    IT_ASSERT(!comm, "communicator is already initialized.");
    ROCM_CALL(infini_rocm_comm_init(rt, name.c_str(), worldSize, rank));
    comm = std::make_unique<RcclCommunicatorObj>(worldSize, rank);
    dropPlans(); // the row-parallel overlap rule depends on the world size
}

void RocmRuntimeObj::savePerfData(const string &path) { PerfEngine::getInstance().savePerfEngineData(path); }
size_t RocmRuntimeObj::perfEpoch = 0;
void RocmRuntimeObj::loadPerfData(const string &path) {
    PerfEngine::getInstance().loadPerfEngineData(path);
    ++perfEpoch;
}
void RocmRuntimeObj::clearPerfData() {
    PerfEngine::getInstance().set_data({});
    ++perfEpoch;
}
size_t RocmRuntimeObj::perfDataSize() { return PerfEngine::getInstance().get_data().size(); }

CommunicatorObj &RocmRuntimeObj::getCommunicator() const {
    IT_ASSERT(comm != nullptr, "communicator is not initialized (call init_comm)");
    return *comm;
}

// ---- the hot loop: the launch plan (rocm_fusion.cc), item by item, asynchronously on the runtime stream -----
void RocmRuntimeObj::launchAll(const Graph &graph, bool validate) const {
    IT_ASSERT(graph != nullptr, "Cannot run a null graph");
    if (validate)
        graph->validateMemory();
    executePlan(*planOf(graph), graph->getOperators());
}

std::shared_ptr<const RocmRuntimeObj::LaunchPlan> RocmRuntimeObj::planOf(const Graph &graph) const {
    const size_t generation = graph->getCaptureGeneration(), perfNow = perfEpoch;
    for (auto it = plans.begin(); it != plans.end();) // graphs that are gone
        it = it->second.owner.expired() ? plans.erase(it) : std::next(it);
    auto &e = plans[graph.get()];
    auto owner = e.owner.lock();
    if (!(e.plan && owner && owner.get() == graph.get() && e.generation == generation && e.perfEpoch == perfNow && e.fusion == fusion)) {
        e.plan.reset(); // (its closures hold the graph's tensors)
        e.owner = WRef<GraphObj>(graph);
        e.graphId = graph->getCaptureStateId();
        e.generation = generation;
        e.perfEpoch = perfNow;
        e.fusion = fusion;
        e.plan = std::make_shared<const LaunchPlan>(buildPlan(graph));
    }
    return e.plan;
}

void RocmRuntimeObj::dropPlans() const { plans.clear(); }

void RocmRuntimeObj::executePlan(const LaunchPlan &plan, const OpVec &ops) const {
    struct ForwardScope { // the plan's forwarded tensors resolve for the duration of its launches only
        ForwardScope(const ForwardMap *m) { RocmRuntimeObj::forwards = m; }
        ~ForwardScope() { RocmRuntimeObj::forwards = nullptr; }
    } scope(plan.forwarded && !plan.forwarded->empty() ? plan.forwarded.get() : nullptr);
    for (const auto &item : plan.items) {
        if (!item.run)
            continue;
        try {
            item.run();
        } catch (Exception &e) {
            if (item.fused)
                e << " while launching (fused: " << item.what << ") " << ops[item.members.front()]->toString();
            throw;
        }
        if (item.fused)
            ++fusedCount;
    }
}

void RocmRuntimeObj::launchOne(const Operator &op) const {
    auto attrs = KernelAttrs{device, op->getOpType().underlying()};
    Kernel *kernel = KernelRegistry::getInstance().getKernel(attrs);
    auto perfKey = PerfEngine::Key{attrs, op->getOpPerfKey()};
    auto perfData = PerfEngine::getInstance().getPerfData(perfKey);
    try {
        if (perfData)
            kernel->getComputeFunc(perfKey)(op, perfData, this);
        else
            kernel->compute(op, this);
    } catch (Exception &e) {
        e << " while launching " << op->toString();
        throw;
    }
}

void RocmRuntimeObj::runWithoutSync(const Graph &graph) const {
    std::lock_guard<std::recursive_mutex> lock(executionMutex);
    launchAll(graph, true);
}

void RocmRuntimeObj::tuneImpl(const Graph &graph, bool profiling) const {
    IT_ASSERT(graph != nullptr, "Cannot tune a null graph");
    graph->validateMemory();
    const auto &registry = KernelRegistry::getInstance();
    auto &perfEngine = PerfEngine::getInstance();
    double total = 0;
    std::map<OpType, double> opTime;
    std::map<OpType, int> opCnt;
    for (auto &op : graph->getOperators()) {
        auto attrs = KernelAttrs{device, op->getOpType().underlying()};
        Kernel *kernel = registry.getKernel(attrs);
        auto perfKey = PerfEngine::Key{attrs, op->getOpPerfKey()};
        PerfRecord record = perfEngine.getPerfData(perfKey);
        if (!record) {
            record = kernel->tune(op, this);
            perfEngine.setPerfData(perfKey, record);
        }
        total += record->time;
        kernel->computeFuncTune(perfKey, op, record, this);
        if (profiling) {
            double t = timeit([&]() { kernel->getComputeFunc(perfKey)(op, record, this); }, [&]() { sync(); }, 1, 1);
            op->print();
            printf(" op_time on rocm %lf\n", t);
            opTime[op->getOpType()] += t;
            opCnt[op->getOpType()]++;
            total += t;
        }
    }
    if (profiling)
        printProfilingData(total, opTime, opCnt);
    ++perfEpoch;
}

void RocmRuntimeObj::run(const Graph &graph, bool tune, bool profiling) const {
    std::lock_guard<std::recursive_mutex> lock(executionMutex);
    if (tune || profiling) {
        tuneImpl(graph, profiling);
        sync();
        return;
    }
    launchAll(graph, true);
    sync(); // host blocks once per graph (reference: syncImpl, cuda_runtime.cc:481-483)
}

// ---- hipGraph capture / replay cache -------------------------------------------------------------
RocmRuntimeObj::CacheEntry::~CacheEntry() {
    if (graph)
        infini_rocm_graph_destroy(graph);
}

RocmRuntimeObj::GraphState RocmRuntimeObj::stateOf(const Graph &graph) const {
    GraphState st{graph->getCaptureStateId(), graph->getTopologyEpoch(), {}};
    st.tensors.reserve(graph->getTensors().size());
    for (const auto &t : graph->getTensors()) {
        const auto &blob = t->getDataBlob();
        IT_ASSERT(blob != nullptr, "Cannot capture a Tensor without memory");
        st.tensors.push_back(TensorState{t.get(), t->getDTypeIndex(), t->getDims(), blob->getStorageId(),
                                         blob->getStorageOffset(), t->getRawDataPtr<const void *>()});
    }
    return st;
}

void RocmRuntimeObj::replay(CacheEntry &entry) {
    try {
        ROCM_CALL(infini_rocm_graph_launch(rt, entry.graph));
        sync();
    } catch (...) {
        // a failed replay poisons nothing but this entry: drop every capture of that graph
        const uint64_t id = entry.state.graphId;
        cache.remove_if([id](const std::unique_ptr<CacheEntry> &e) { return e->state.graphId == id; });
        infini_rocm_graph_abort_capture(rt);
        throw;
    }
}

void RocmRuntimeObj::runWithHipGraph(const Graph &graph) {
    IT_ASSERT(graph != nullptr, "Cannot run a null graph");
    std::lock_guard<std::recursive_mutex> executionLock(executionMutex);
    std::lock_guard<std::recursive_mutex> cacheLock(cacheMutex);
    // drop captures whose graph object is gone
    cache.remove_if([](const std::unique_ptr<CacheEntry> &e) { return e->owner.expired(); });

    const size_t generation = graph->getCaptureGeneration();
    // fast path: the most recent capture of this very graph at this generation (nothing about its
    // topology, shapes or storage changed since — GraphObj bumps the generation whenever it does)
    for (auto it = cache.begin(); it != cache.end(); ++it) {
        auto owner = (*it)->owner.lock();
        if (owner && owner.get() == graph.get() && (*it)->generation == generation) {
            cache.splice(cache.begin(), cache, it);
            replay(*cache.front());
            return;
        }
    }
    graph->validateMemory();
    GraphState state = stateOf(graph);
    for (auto it = cache.begin(); it != cache.end(); ++it) {
        auto owner = (*it)->owner.lock();
        if (owner && owner.get() == graph.get() && (*it)->state == state) {
            (*it)->generation = generation;
            cache.splice(cache.begin(), cache, it);
            replay(*cache.front());
            return;
        }
    }
    // capture. Kernels must not allocate or synchronise while the stream records; a kernel that does
    // makes the capture fail, the stream is rebuilt and the error propagates (reference:
    // recoverExecutionStreamAfterFailure, cuda_runtime.cc:226-250; test_cudagraph.cc:18-27).
    auto entry = std::make_unique<CacheEntry>();
    entry->owner = WRef<GraphObj>(graph);
    entry->state = std::move(state);
    entry->generation = generation;
    // Kernels take scratch from the runtime workspace. It may grow while the stream records (the C ABI retires the
    // outgrown block instead of freeing it, so launches already recorded — and every graph exec captured earlier —
    // keep valid addresses). The graph is never executed eagerly on behalf of a capture: a graph with collectives
    // must issue the same RCCL calls on every rank whatever the state of each rank's workspace. Only if a capture
    // fails AND the workspace block changed during it (a HIP build that refuses allocation under capture) is the
    // capture repeated once, now with a workspace that is large enough; launches recorded into an aborted capture
    // never execute.
    // The plan is made BEFORE the stream records: the planner may read one-element constants back from the device
    // (a decomposed Gelu's sqrt 2, LayerNorm's epsilon), which a recording stream cannot serve.
    const std::shared_ptr<const LaunchPlan> planRef = planOf(graph);
    const LaunchPlan &plan = *planRef;
    for (int attempt = 0;; ++attempt) {
        uint64_t epochBefore = 0, epochAfter = 0;
        ROCM_CALL(infini_rocm_workspace_info(rt, nullptr, nullptr, &epochBefore));
        ROCM_CALL(infini_rocm_graph_begin_capture(rt));
        try {
            executePlan(plan, graph->getOperators());
            ROCM_CALL(infini_rocm_graph_end_capture(rt, &entry->graph));
            break;
        } catch (...) {
            infini_rocm_graph_abort_capture(rt);
            infini_rocm_workspace_info(rt, nullptr, nullptr, &epochAfter);
            if (attempt > 0 || epochAfter == epochBefore)
                throw;
        }
    }
    IT_ASSERT(generation == graph->getCaptureGeneration(), "Graph changed while hipGraph capture was in progress");
    ROCM_CALL(infini_rocm_graph_launch(rt, entry->graph));
    sync();
    ++captureCount;
    cache.push_front(std::move(entry));
    while (cache.size() > cacheCapacity)
        cache.pop_back();
}

void RocmRuntimeObj::invalidateGraphCaptureCache(uint64_t graphId) noexcept {
    std::lock_guard<std::recursive_mutex> executionLock(executionMutex);
    std::lock_guard<std::recursive_mutex> cacheLock(cacheMutex);
    cache.remove_if([graphId](const std::unique_ptr<CacheEntry> &e) { return e->state.graphId == graphId; });
    for (auto it = plans.begin(); it != plans.end();) // the graph changed (or died): its plan holds its tensors
        it = it->second.graphId == graphId ? plans.erase(it) : std::next(it);
}

void RocmRuntimeObj::clearHipGraphCache() {
    std::lock_guard<std::recursive_mutex> executionLock(executionMutex);
    std::lock_guard<std::recursive_mutex> cacheLock(cacheMutex);
    cache.clear();
    // no graph exec of this runtime is alive any more: scratch blocks retired by workspace growth can go
    ROCM_CALL(infini_rocm_workspace_trim(rt));
}
size_t RocmRuntimeObj::getHipGraphCacheSize() const {
    std::lock_guard<std::recursive_mutex> lock(cacheMutex);
    return cache.size();
}
size_t RocmRuntimeObj::getHipGraphCaptureCount() const {
    std::lock_guard<std::recursive_mutex> lock(cacheMutex);
    return captureCount;
}

} // namespace infini
