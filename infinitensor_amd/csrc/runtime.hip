// Runtime part of the C ABI: device, stream, memory, workspace, events, hipGraph capture.
// Mirrors what CudaRuntimeObj owns in the reference (src/cuda/cuda_runtime.cc:30-120, 252-426,
// 481-493) but as a plain C handle; the C++ plugin's RocmRuntimeObj wraps one of these.
#include "common.h"

namespace irocm {
static thread_local char g_err[1024] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
} // namespace irocm

namespace irocm {
const void *wcache_lookup(infiniRocmRuntime *rt, const void *src, int f, int c, int rs, int kind) {
    for (const auto &e : rt->wcache)
        if (e.src == src && e.f == f && e.c == c && e.rs == rs && e.kind == kind)
            return e.packed;
    return nullptr;
}
int wcache_insert(infiniRocmRuntime *rt, const void *src, size_t src_bytes, int f, int c, int rs, int kind, size_t packed_bytes,
                  void **packed, hipStream_t *stream) {
    IROCM_HIP(hipSetDevice(rt->device));
    void *buf = nullptr;
    hipError_t e;
    if (rt->capturing) { // allocation and the side stream are legal under a thread-local Relaxed capture mode
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        IROCM_HIP(hipThreadExchangeStreamCaptureMode(&mode));
        e = hipMalloc(&buf, packed_bytes);
        if (e == hipSuccess && !rt->side_stream)
            e = hipStreamCreateWithFlags(&rt->side_stream, hipStreamNonBlocking);
        (void)hipThreadExchangeStreamCaptureMode(&mode);
    } else {
        e = hipMalloc(&buf, packed_bytes);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (buf)
            (void)hipFree(buf);
        IROCM_FAIL(e == hipErrorOutOfMemory ? INFINI_ROCM_OUT_OF_MEMORY : INFINI_ROCM_HIP_ERROR,
                   "weight cache: allocation of %zu bytes failed: %s", packed_bytes, hipGetErrorString(e));
    }
    rt->wcache.push_back({src, src_bytes, f, c, rs, kind, buf, packed_bytes});
    *packed = buf;
    *stream = rt->capturing ? rt->side_stream : rt->stream;
    return INFINI_ROCM_OK;
}
int wcache_commit(infiniRocmRuntime *rt, hipStream_t stream) {
    if (stream == rt->stream)
        return INFINI_ROCM_OK; // same stream as the consumer: ordered
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    IROCM_HIP(hipThreadExchangeStreamCaptureMode(&mode));
    const hipError_t e = hipStreamSynchronize(stream); // the captured graph must find the image complete at every replay
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    if (e != hipSuccess)
        IROCM_FAIL(INFINI_ROCM_HIP_ERROR, "weight cache: side stream failed: %s", hipGetErrorString(e));
    return INFINI_ROCM_OK;
}
void wcache_forget(infiniRocmRuntime *rt, const void *packed) {
    for (size_t i = 0; i < rt->wcache.size(); ++i)
        if (rt->wcache[i].packed == packed) {
            rt->retired.push_back(rt->wcache[i].packed);
            rt->wcache.erase(rt->wcache.begin() + i);
            ++rt->wcache_epoch;
            return;
        }
}
void wcache_invalidate(infiniRocmRuntime *rt, const void *ptr, size_t bytes) {
    if (rt->wcache.empty() || !ptr || !bytes)
        return;
    const char *lo = (const char *)ptr, *hi = lo + bytes;
    for (size_t i = 0; i < rt->wcache.size();) {
        const auto &e = rt->wcache[i];
        const char *slo = (const char *)e.src, *shi = slo + e.src_bytes;
        if (slo < hi && lo < shi) {
            rt->retired.push_back(e.packed); // a captured graph may still read it: released with the retired workspace blocks
            rt->wcache.erase(rt->wcache.begin() + i);
            ++rt->wcache_epoch;
        } else {
            ++i;
        }
    }
}
} // namespace irocm

using namespace irocm;

extern "C" int infini_rocm_comm_destroy(infiniRocmRuntime_t rt);

extern "C" {

const char *infini_rocm_last_error(void) { return irocm::g_err; }

const char *infini_rocm_version(void) { return "infinitensor_amd 0.1 (gfx950)"; }

int infini_rocm_device_count(int *count) {
    IROCM_CHECK_ARG(count, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return INFINI_ROCM_OK;
}

int infini_rocm_runtime_create(int device, infiniRocmRuntime_t *out) {
    IROCM_CHECK_ARG(out, "out is NULL");
    int n = 0;
    IROCM_HIP(hipGetDeviceCount(&n));
    IROCM_CHECK_ARG(device >= 0 && device < n, "device %d out of range (have %d)", device, n);
    IROCM_HIP(hipSetDevice(device));
    auto *rt = new infiniRocmRuntime();
    rt->device = device;
    hipError_t e = hipStreamCreateWithFlags(&rt->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete rt;
        IROCM_FAIL(INFINI_ROCM_HIP_ERROR, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    rt->stream = rt->own_stream;
    if (hipMalloc(&rt->zeros, 256) != hipSuccess || hipMemset(rt->zeros, 0, 256) != hipSuccess ||
        hipMalloc((void **)&rt->sync_flags, infiniRocmRuntime::kSyncFlagWords * 4) != hipSuccess ||
        hipMemset(rt->sync_flags, 0, infiniRocmRuntime::kSyncFlagWords * 4) != hipSuccess ||
        hipHostMalloc((void **)&rt->sync_err_host, 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&rt->sync_err_dev, (void *)rt->sync_err_host, 0) != hipSuccess) {
        (void)hipStreamDestroy(rt->own_stream);
        delete rt;
        IROCM_FAIL(INFINI_ROCM_HIP_ERROR, "cannot allocate the runtime's zero block");
    }
    *rt->sync_err_host = 0u;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess)
        rt->num_cu = prop.multiProcessorCount;
    *out = rt;
    return INFINI_ROCM_OK;
}

int infini_rocm_runtime_destroy(infiniRocmRuntime_t rt) {
    if (!rt)
        return INFINI_ROCM_OK;
    (void)hipSetDevice(rt->device);
    if (rt->own_stream) {
        (void)hipStreamSynchronize(rt->own_stream);
    }
    if (rt->comm_stream)
        (void)hipStreamSynchronize(rt->comm_stream);
    (void)infini_rocm_comm_destroy(rt);
    for (hipEvent_t e : rt->comm_events)
        (void)hipEventDestroy(e);
    if (rt->comm_stream)
        (void)hipStreamDestroy(rt->comm_stream);
    if (rt->workspace)
        (void)hipFree(rt->workspace);
    for (void *p : rt->retired)
        (void)hipFree(p);
    for (auto &e : rt->wcache)
        (void)hipFree(e.packed);
    if (rt->side_stream)
        (void)hipStreamDestroy(rt->side_stream);
    if (rt->zeros)
        (void)hipFree(rt->zeros);
    if (rt->sync_flags)
        (void)hipFree(rt->sync_flags);
    if (rt->sync_err_host)
        (void)hipHostFree((void *)rt->sync_err_host);
    if (rt->own_stream)
        (void)hipStreamDestroy(rt->own_stream);
    delete rt;
    return INFINI_ROCM_OK;
}

int infini_rocm_runtime_device_info(infiniRocmRuntime_t rt, infiniRocmDeviceInfo *info) {
    IROCM_CHECK_ARG(rt && info, "NULL argument");
    hipDeviceProp_t prop;
    IROCM_HIP(hipGetDeviceProperties(&prop, rt->device));
    memset(info, 0, sizeof(*info));
    snprintf(info->name, sizeof(info->name), "%s", prop.name);
    snprintf(info->arch, sizeof(info->arch), "%s", prop.gcnArchName);
    info->compute_units = prop.multiProcessorCount;
    info->clock_mhz = prop.clockRate / 1000;
    info->memory_clock_mhz = prop.memoryClockRate / 1000;
    info->memory_bus_bits = prop.memoryBusWidth;
    info->total_memory = prop.totalGlobalMem;
    info->wavefront_size = prop.warpSize;
    info->lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    return INFINI_ROCM_OK;
}

int infini_rocm_runtime_get_stream(infiniRocmRuntime_t rt, void **stream) {
    IROCM_CHECK_ARG(rt && stream, "NULL argument");
    *stream = (void *)rt->stream;
    return INFINI_ROCM_OK;
}

int infini_rocm_runtime_set_stream(infiniRocmRuntime_t rt, void *stream) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(!rt->capturing, "cannot change stream while capturing");
    rt->stream = (hipStream_t)stream;
    return INFINI_ROCM_OK;
}

int infini_rocm_runtime_use_own_stream(infiniRocmRuntime_t rt) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(!rt->capturing, "cannot change stream while capturing");
    rt->stream = rt->own_stream;
    return INFINI_ROCM_OK;
}

int infini_rocm_runtime_sync(infiniRocmRuntime_t rt) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_HIP(hipStreamSynchronize(rt->stream));
    if (rt->comm_stream && rt->comm_pending) // async collectives nobody joined yet (comm.hip)
        IROCM_HIP(hipStreamSynchronize(rt->comm_stream));
    // the hand-written transport's kernels give up after a time limit instead of hanging and leave an error word: the first sync
    // behind such a collective reports it (RCCL error code) — nothing else would, and the tensors are garbage (round-4 advisor)
    // in-launch exchanges (split-K of the conv tap GEMM): a wave whose partner slice did not arrive within the time limit left this
    // word set; the sums of that launch are wrong and a late producer may have left a flag set — re-zero them all, report once
    if (rt->sync_err_host && *rt->sync_err_host) {
        *rt->sync_err_host = 0u;
        IROCM_HIP(hipMemsetAsync(rt->sync_flags, 0, infiniRocmRuntime::kSyncFlagWords * 4, rt->stream));
        IROCM_HIP(hipStreamSynchronize(rt->stream));
        if (rt->dcomm && rt->dcomm_dirty)
            (void)irocm::direct_check(rt);
        IROCM_FAIL(INFINI_ROCM_HIP_ERROR, "split-K exchange: a partner workgroup did not arrive within the time limit (the outputs of the "
                                          "launches since the last sync are undefined; flags re-zeroed, error cleared). Kernels spinning on "
                                          "another stream can cause this: do not overlap direct-transport collectives with split-K convolutions");
    }
    if (rt->dcomm && rt->dcomm_dirty)
        return irocm::direct_check(rt);
    return INFINI_ROCM_OK;
}

static constexpr size_t kAllocSlack = 256;

int infini_rocm_alloc(infiniRocmRuntime_t rt, size_t bytes, void **ptr) {
    IROCM_CHECK_ARG(rt && ptr, "NULL argument");
    IROCM_HIP(hipSetDevice(rt->device));
    *ptr = nullptr;
    if (bytes == 0)
        return INFINI_ROCM_OK;
    // 256 bytes of slack behind AND in front of every block: kernels that fetch whole 16-byte runs (the conv mode of the persistent GEMM
    // on planes that are not a multiple of 8 pixels) may read a few bytes past the last tensor of an arena, and the tap mode (3 x 3
    // layers: a tap moves a run by up to one image row + one pixel) the same distance in front of the first one. The slack is never
    // written and what is read from it is masked away. infini_rocm_dealloc undoes the offset.
    void *raw = nullptr;
    IROCM_HIP(hipMalloc(&raw, bytes + 2 * kAllocSlack));
    *ptr = (char *)raw + kAllocSlack;
    return INFINI_ROCM_OK;
}

int infini_rocm_dealloc(infiniRocmRuntime_t rt, void *ptr) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    if (ptr) {
        if (!rt->wcache.empty()) { // the block's size is not known here: drop every image packed from inside it
            hipDeviceptr_t base = nullptr;
            size_t size = 0;
            if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr) == hipSuccess)
                wcache_invalidate(rt, base, size);
            else
                (void)hipGetLastError();
        }
        IROCM_HIP(hipFree((char *)ptr - kAllocSlack));
    }
    return INFINI_ROCM_OK;
}

// Host<->device copies are ordered with the runtime stream and block the caller, like the
// reference's cudaMemcpy-based copyBlobFromCPU/ToCPU (src/cuda/cuda_runtime.cc:485-493).
int infini_rocm_copy_from_cpu(infiniRocmRuntime_t rt, void *dst, const void *src, size_t bytes) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    if (bytes == 0)
        return INFINI_ROCM_OK;
    wcache_invalidate(rt, dst, bytes);
    IROCM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, rt->stream));
    IROCM_HIP(hipStreamSynchronize(rt->stream));
    return INFINI_ROCM_OK;
}

int infini_rocm_copy_to_cpu(infiniRocmRuntime_t rt, void *dst, const void *src, size_t bytes) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    if (bytes == 0)
        return INFINI_ROCM_OK;
    IROCM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, rt->stream));
    IROCM_HIP(hipStreamSynchronize(rt->stream));
    return INFINI_ROCM_OK;
}

int infini_rocm_copy_inside(infiniRocmRuntime_t rt, void *dst, const void *src, size_t bytes) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    if (bytes == 0 || dst == src)
        return INFINI_ROCM_OK;
    wcache_invalidate(rt, dst, bytes);
    IROCM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, rt->stream));
    return INFINI_ROCM_OK;
}

int infini_rocm_memset(infiniRocmRuntime_t rt, void *dst, int value, size_t bytes) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    if (bytes == 0)
        return INFINI_ROCM_OK;
    wcache_invalidate(rt, dst, bytes);
    IROCM_HIP(hipMemsetAsync(dst, value, bytes, rt->stream));
    return INFINI_ROCM_OK;
}

// The scratch block never moves under a captured graph: growing allocates a NEW block and RETIRES the old one
// (kept allocated — hipGraph execs captured earlier have its address baked into their kernel nodes; the reference
// avoids the problem with one fixed 7 GiB block, cuda_runtime.h:85-88). Growth is legal during stream capture:
// hipMalloc is issued under a thread-local Relaxed capture mode (the same guard torch's caching allocator uses), and
// the launches already recorded keep pointing at the retired block, which stays valid. Retired blocks are released by
// infini_rocm_workspace_trim (the plugin calls it when its graph cache is empty) and at runtime destruction.
int infini_rocm_workspace(infiniRocmRuntime_t rt, size_t bytes, void **ptr) {
    IROCM_CHECK_ARG(rt && ptr, "NULL argument");
    if (bytes > rt->workspace_bytes) {
        IROCM_HIP(hipSetDevice(rt->device));
        // geometric growth bounds the retired total by the final size
        size_t want = bytes > 2 * rt->workspace_bytes ? bytes : 2 * rt->workspace_bytes;
        want = (want + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
        void *fresh = nullptr;
        hipError_t e;
        if (rt->capturing) {
            hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
            IROCM_HIP(hipThreadExchangeStreamCaptureMode(&mode));
            e = hipMalloc(&fresh, want);
            (void)hipThreadExchangeStreamCaptureMode(&mode);
        } else {
            e = hipMalloc(&fresh, want);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            IROCM_FAIL(e == hipErrorOutOfMemory ? INFINI_ROCM_OUT_OF_MEMORY : INFINI_ROCM_HIP_ERROR,
                       "workspace: hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        }
        if (rt->workspace)
            rt->retired.push_back(rt->workspace);
        rt->workspace = fresh;
        rt->workspace_bytes = want;
        ++rt->workspace_epoch;
    }
    *ptr = rt->workspace;
    return INFINI_ROCM_OK;
}

int infini_rocm_workspace_trim(infiniRocmRuntime_t rt) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(!rt->capturing, "cannot trim the workspace while capturing");
    if (rt->retired.empty())
        return INFINI_ROCM_OK;
    IROCM_HIP(hipSetDevice(rt->device));
    IROCM_HIP(hipStreamSynchronize(rt->stream));
    for (void *p : rt->retired)
        (void)hipFree(p);
    rt->retired.clear();
    return INFINI_ROCM_OK;
}

int infini_rocm_conv2d_set_const_weights(infiniRocmRuntime_t rt, int on) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    rt->conv_const_weights = on ? 1 : 0;
    return INFINI_ROCM_OK;
}

int infini_rocm_weight_cache_info(infiniRocmRuntime_t rt, size_t *entries, size_t *bytes, uint64_t *epoch) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    size_t b = 0;
    for (const auto &e : rt->wcache)
        b += e.packed_bytes;
    if (entries) *entries = rt->wcache.size();
    if (bytes) *bytes = b;
    if (epoch) *epoch = rt->wcache_epoch;
    return INFINI_ROCM_OK;
}

int infini_rocm_weight_cache_clear(infiniRocmRuntime_t rt) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    for (auto &e : rt->wcache) {
        rt->retired.push_back(e.packed);
        ++rt->wcache_epoch;
    }
    rt->wcache.clear();
    return INFINI_ROCM_OK;
}

int infini_rocm_workspace_info(infiniRocmRuntime_t rt, size_t *bytes, size_t *retired_blocks, uint64_t *epoch) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    if (bytes) *bytes = rt->workspace_bytes;
    if (retired_blocks) *retired_blocks = rt->retired.size();
    if (epoch) *epoch = rt->workspace_epoch;
    return INFINI_ROCM_OK;
}

// ---- events --------------------------------------------------------------------------------
int infini_rocm_event_create(infiniRocmEvent_t *ev) {
    IROCM_CHECK_ARG(ev, "NULL argument");
    auto *e = new infiniRocmEvent();
    hipError_t r = hipEventCreate(&e->ev);
    if (r != hipSuccess) {
        delete e;
        IROCM_FAIL(INFINI_ROCM_HIP_ERROR, "hipEventCreate failed: %s", hipGetErrorString(r));
    }
    *ev = e;
    return INFINI_ROCM_OK;
}

int infini_rocm_event_destroy(infiniRocmEvent_t ev) {
    if (ev) {
        (void)hipEventDestroy(ev->ev);
        delete ev;
    }
    return INFINI_ROCM_OK;
}

int infini_rocm_event_record(infiniRocmRuntime_t rt, infiniRocmEvent_t ev) {
    IROCM_CHECK_ARG(rt && ev, "NULL argument");
    IROCM_HIP(hipEventRecord(ev->ev, rt->stream));
    return INFINI_ROCM_OK;
}

int infini_rocm_event_elapsed_ms(infiniRocmEvent_t start, infiniRocmEvent_t stop, float *ms) {
    IROCM_CHECK_ARG(start && stop && ms, "NULL argument");
    IROCM_HIP(hipEventSynchronize(stop->ev));
    IROCM_HIP(hipEventElapsedTime(ms, start->ev, stop->ev));
    return INFINI_ROCM_OK;
}

// ---- hipGraph capture ----------------------------------------------------------------------
int infini_rocm_graph_begin_capture(infiniRocmRuntime_t rt) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(!rt->capturing, "capture already active");
    // Work already queued on the stream (asynchronous memset / copy_inside of a raw C-ABI caller) must be complete before
    // anything recorded here runs on the side stream (the conv weight pack of a capture is NOT ordered after the stream).
    IROCM_HIP(hipStreamSynchronize(rt->stream));
    // ThreadLocal mode, like the reference (cuda_runtime.cc:259-266): other threads/runtimes
    // may keep using the device while this stream records.
    IROCM_HIP(hipStreamBeginCapture(rt->stream, hipStreamCaptureModeThreadLocal));
    rt->capturing = true;
    return INFINI_ROCM_OK;
}

int infini_rocm_graph_end_capture(infiniRocmRuntime_t rt, infiniRocmGraph_t *graph) {
    IROCM_CHECK_ARG(rt && graph, "NULL argument");
    IROCM_CHECK_ARG(rt->capturing, "no capture active");
    rt->capturing = false;
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(rt->stream, &g);
    if (e != hipSuccess || !g) {
        (void)hipGetLastError();
        IROCM_FAIL(INFINI_ROCM_CAPTURE_ERROR, "hipStreamEndCapture failed: %s",
                   hipGetErrorString(e));
    }
    hipGraphExec_t x = nullptr;
    e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        IROCM_FAIL(INFINI_ROCM_CAPTURE_ERROR, "hipGraphInstantiate failed: %s",
                   hipGetErrorString(e));
    }
    auto *out = new infiniRocmGraph();
    out->graph = g;
    out->exec = x;
    *graph = out;
    return INFINI_ROCM_OK;
}

// Abandon a capture after a failure inside it (reference: recoverExecutionStreamAfterFailure,
// cuda_runtime.cc:226-250 — the poisoned stream is destroyed and recreated).
int infini_rocm_graph_abort_capture(infiniRocmRuntime_t rt) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    if (rt->capturing) {
        hipGraph_t g = nullptr;
        (void)hipStreamEndCapture(rt->stream, &g);
        if (g)
            (void)hipGraphDestroy(g);
        rt->capturing = false;
    }
    (void)hipGetLastError();
    if (rt->stream == rt->own_stream) {
        (void)hipStreamDestroy(rt->own_stream);
        rt->own_stream = nullptr;
        IROCM_HIP(hipStreamCreateWithFlags(&rt->own_stream, hipStreamNonBlocking));
        rt->stream = rt->own_stream;
    }
    return INFINI_ROCM_OK;
}

int infini_rocm_graph_launch(infiniRocmRuntime_t rt, infiniRocmGraph_t graph) {
    IROCM_CHECK_ARG(rt && graph && graph->exec, "NULL argument");
    IROCM_HIP(hipGraphLaunch(graph->exec, rt->stream));
    return INFINI_ROCM_OK;
}

int infini_rocm_graph_destroy(infiniRocmGraph_t graph) {
    if (graph) {
        if (graph->exec)
            (void)hipGraphExecDestroy(graph->exec);
        if (graph->graph)
            (void)hipGraphDestroy(graph->graph);
        delete graph;
    }
    return INFINI_ROCM_OK;
}

} // extern "C"
