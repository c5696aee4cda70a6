// RCCL communicator and collective operators over xGMI.
//
// Replaces NcclCommunicatorObj (reference: include/cuda/nccl_communicator.h:22-68),
// CudaRuntimeObj::initComm (src/cuda/cuda_runtime.cc:495-509) and the collective kernels
// AllReduceNCCL src/kernels/cuda/all_reduce.cc:8-63, AllGatherNCCL all_gather.cc:8-40,
// BroadcastNCCL broadcast.cc:8-26, SendNCCL send.cc:8-37, RecvNCCL recv.cc:8-41.
//
// One communicator per runtime, one process per GPU. Differences from the reference, on purpose:
//  - every collective is enqueued on the runtime's stream (the reference puts AllGather / Broadcast /
//    Send / Recv on stream 0: all_gather.cc:30, broadcast.cc:23) so it orders with the kernels around
//    it and is capturable in a hipGraph;
//  - RCCL errors are returned (INFINI_ROCM_RCCL_ERROR) instead of exit(EXIT_FAILURE);
//  - besides the file-based id exchange of the reference (kept, same file name scheme, for drop-in
//    `init_comm(name, world, rank)`), the unique id can be passed as bytes so a launcher that already
//    has a control channel (torch.distributed / multiprocessing) skips the filesystem;
//  - f16, bf16, f32, f64, i8, u8, i32, i64 payloads (reference: f32/f16/i8 for all-reduce, f32 only for
//    the others).
// Topology note (MI355X): 8 GPUs fully connected by xGMI, 7 links x ~153 GB/s per GPU. RCCL picks the
// algorithm; a message is the whole activation tensor (two per transformer block under Megatron TP).
#include "common.h"

#include <chrono>
#include <cstdlib>
#include <fstream>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

namespace irocm {

#define IROCM_NCCL(expr)                                                                           \
    do {                                                                                           \
        ncclResult_t _r = (expr);                                                                  \
        if (_r != ncclSuccess)                                                                     \
            IROCM_FAIL(INFINI_ROCM_RCCL_ERROR, "%s failed: %s", #expr, ncclGetErrorString(_r));    \
    } while (0)

static bool nccl_type(int dtype, ncclDataType_t *t) {
    switch (dtype) {
    case INFINI_DT_F32: *t = ncclFloat32; return true;
    case INFINI_DT_F16: *t = ncclFloat16; return true;
    case INFINI_DT_BF16: *t = ncclBfloat16; return true;
    case INFINI_DT_F64: *t = ncclFloat64; return true;
    case INFINI_DT_I8: *t = ncclInt8; return true;
    case INFINI_DT_U8: case INFINI_DT_BOOL: *t = ncclUint8; return true;
    case INFINI_DT_I32: *t = ncclInt32; return true;
    case INFINI_DT_U32: *t = ncclUint32; return true;
    case INFINI_DT_I64: *t = ncclInt64; return true;
    case INFINI_DT_U64: *t = ncclUint64; return true;
    default: return false;
    }
}

static int comm_init_with_id(infiniRocmRuntime_t rt, const ncclUniqueId &id, int world, int rank) {
    IROCM_CHECK_ARG(rt->comm == nullptr, "communicator already initialised");
    IROCM_HIP(hipSetDevice(rt->device));
    ncclComm_t comm = nullptr;
    IROCM_NCCL(ncclCommInitRank(&comm, world, id, rank));
    rt->comm = (void *)comm;
    rt->comm_world = world;
    rt->comm_rank = rank;
    return INFINI_ROCM_OK;
}

// fork: `to` waits for everything enqueued on `from` so far. Works under stream capture too (the event record / wait
// become an edge of the captured graph and pull `to` into the capture; comm_join brings it back before end_capture).
static int stream_fork(infiniRocmRuntime_t rt, hipStream_t from, hipStream_t to) {
    if (rt->comm_events.size() < 16) {
        hipEvent_t e = nullptr;
        IROCM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        rt->comm_events.push_back(e);
        rt->comm_event_next = rt->comm_events.size() - 1;
    }
    hipEvent_t ev = rt->comm_events[rt->comm_event_next];
    rt->comm_event_next = (rt->comm_event_next + 1) % rt->comm_events.size();
    IROCM_HIP(hipEventRecord(ev, from));
    IROCM_HIP(hipStreamWaitEvent(to, ev, 0));
    return INFINI_ROCM_OK;
}

static int ensure_comm_stream(infiniRocmRuntime_t rt) {
    if (!rt->comm_stream) {
        IROCM_HIP(hipSetDevice(rt->device));
        IROCM_HIP(hipStreamCreateWithFlags(&rt->comm_stream, hipStreamNonBlocking));
    }
    return INFINI_ROCM_OK;
}

// y[i] = sum over the `parts` slabs of x (x: [parts][count], y: [count]); fp32 accumulation for 16-bit types
template <typename T> __global__ __launch_bounds__(256) void sum_slabs_kernel(const T *x, T *y, long count, int parts) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) {
        float acc = 0.f;
        for (int p = 0; p < parts; ++p)
            acc += (float)x[(long)p * count + i];
        y[i] = (T)acc;
    }
}

// the hand-written transport serves the call when it is the only one initialised or was selected (comm_set_algo)
static bool use_direct(const infiniRocmRuntime *rt) { return rt->dcomm && (rt->comm_algo == 1 || !rt->comm); }
// Synchronous direct-transport collectives run on the runtime stream, the *_async ones on the comm stream, and both families share
// the communicator's sequence number, flag words and parity boxes: a synchronous one issued before comm_join must not run
// beside a pending async one (RCCL serialises this case itself). The runtime stream joins the comm stream first.
static int direct_order(infiniRocmRuntime *rt) {
    if (rt->comm_pending && rt->comm_stream)
        return infini_rocm_comm_join(rt);
    return INFINI_ROCM_OK;
}
static bool direct_wanted_by_env() {
    const char *e = std::getenv("INFINI_ROCM_COMM");
    return e && std::string(e) == "direct";
}

} // namespace irocm

using namespace irocm;

extern "C" {

int infini_rocm_comm_unique_id(void *buf, size_t *nbytes) {
    IROCM_CHECK_ARG(buf && nbytes && *nbytes >= sizeof(ncclUniqueId), "buffer too small (need %zu)",
                    sizeof(ncclUniqueId));
    ncclUniqueId id;
    IROCM_NCCL(ncclGetUniqueId(&id));
    memcpy(buf, &id, sizeof(id));
    *nbytes = sizeof(id);
    return INFINI_ROCM_OK;
}

int infini_rocm_comm_init_id(infiniRocmRuntime_t rt, const void *unique_id, size_t nbytes, int world_size,
                             int rank) {
    IROCM_CHECK_ARG(rt && unique_id, "NULL argument");
    IROCM_CHECK_ARG(nbytes == sizeof(ncclUniqueId), "unique id must be %zu bytes", sizeof(ncclUniqueId));
    IROCM_CHECK_ARG(world_size >= 1 && rank >= 0 && rank < world_size, "bad world/rank %d/%d", world_size, rank);
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    return comm_init_with_id(rt, id, world_size, rank);
}

// File-based rendezvous, same protocol as the reference (nccl_communicator.h:27-51): rank 0 writes
// ./<name>_nccl_id.bin, the others poll for it (100 ms, 10 s limit); rank 0 removes it afterwards.
// The id is written to a temporary name and renamed so that a reader never sees a partial file.
int infini_rocm_comm_init(infiniRocmRuntime_t rt, const char *name, int world_size, int rank) {
    IROCM_CHECK_ARG(rt && name, "NULL argument");
    IROCM_CHECK_ARG(world_size >= 1 && rank >= 0 && rank < world_size, "bad world/rank %d/%d", world_size, rank);
    if (direct_wanted_by_env()) // INFINI_ROCM_COMM=direct: the hand-written IPC / xGMI transport instead of RCCL
        return direct_init(rt, name, world_size, rank);
    const std::string path = std::string("./") + name + "_nccl_id.bin";
    ncclUniqueId id;
    if (rank == 0) {
        IROCM_NCCL(ncclGetUniqueId(&id));
        const std::string tmp = path + ".tmp";
        {
            std::ofstream ofs(tmp, std::ios::binary | std::ios::trunc);
            ofs.write((const char *)&id, sizeof(id));
        }
        if (rename(tmp.c_str(), path.c_str()) != 0)
            IROCM_FAIL(INFINI_ROCM_RCCL_ERROR, "cannot publish %s", path.c_str());
    } else {
        const auto begin = std::chrono::steady_clock::now();
        struct stat st;
        while (stat(path.c_str(), &st) != 0 || (size_t)st.st_size < sizeof(id)) {
            if (std::chrono::steady_clock::now() > begin + std::chrono::seconds(10))
                IROCM_FAIL(INFINI_ROCM_RCCL_ERROR, "time limit (10s) exceeded waiting for %s", path.c_str());
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
        std::ifstream ifs(path, std::ios::binary);
        ifs.read((char *)&id, sizeof(id));
    }
    const int st = comm_init_with_id(rt, id, world_size, rank);
    if (rank == 0)
        (void)unlink(path.c_str());
    return st;
}

// The hand-written transport next to (or instead of) RCCL: same rendezvous scheme, files ./<name>_xgmi_<rank>.bin.
int infini_rocm_comm_init_direct(infiniRocmRuntime_t rt, const char *name, int world_size, int rank) {
    IROCM_CHECK_ARG(rt && name, "NULL argument");
    IROCM_CHECK_ARG(world_size >= 1 && rank >= 0 && rank < world_size, "bad world/rank %d/%d", world_size, rank);
    IROCM_CHECK_ARG(!rt->comm || (rt->comm_world == world_size && rt->comm_rank == rank),
                    "direct transport: world/rank %d/%d differ from the RCCL communicator's %d/%d", world_size, rank, rt->comm_world,
                    rt->comm_rank);
    return direct_init(rt, name, world_size, rank);
}

int infini_rocm_comm_set_algo(infiniRocmRuntime_t rt, int algo) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(algo == 0 || algo == 1, "comm_set_algo: 0 (RCCL when initialised) or 1 (direct)");
    IROCM_CHECK_ARG(algo == 0 || rt->dcomm, "comm_set_algo: the direct transport is not initialised (infini_rocm_comm_init_direct)");
    rt->comm_algo = algo;
    return INFINI_ROCM_OK;
}

int infini_rocm_comm_check(infiniRocmRuntime_t rt) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    return direct_check(rt);
}

int infini_rocm_comm_destroy(infiniRocmRuntime_t rt) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    if (rt->dcomm) {
        (void)direct_destroy(rt);
        rt->comm_algo = 0;
        if (!rt->comm) {
            rt->comm_world = 1;
            rt->comm_rank = 0;
        }
    }
    if (rt->comm) {
        ncclComm_t c = (ncclComm_t)rt->comm;
        (void)hipStreamSynchronize(rt->stream);
        (void)ncclCommDestroy(c);
        rt->comm = nullptr;
        rt->comm_world = 1;
        rt->comm_rank = 0;
    }
    return INFINI_ROCM_OK;
}

int infini_rocm_comm_info(infiniRocmRuntime_t rt, int *world_size, int *rank) {
    IROCM_CHECK_ARG(rt && world_size && rank, "NULL argument");
    IROCM_CHECK_ARG(rt->comm || rt->dcomm, "communicator not initialised");
    *world_size = rt->comm_world;
    *rank = rt->comm_rank;
    return INFINI_ROCM_OK;
}

// op: 0 sum, 1 prod, 2 min, 3 max, 4 avg (reference: AllReduce{Sum,Prod,Min,Max,Avg})
int infini_rocm_all_reduce(infiniRocmRuntime_t rt, int op, int dtype, const void *x, void *y, int64_t count) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(rt->comm || rt->dcomm, "all_reduce: communicator not initialised (call init_comm)");
    static const ncclRedOp_t ops[] = {ncclSum, ncclProd, ncclMin, ncclMax, ncclAvg};
    IROCM_CHECK_ARG(op >= 0 && op <= 4, "all_reduce: bad op %d", op);
    if (use_direct(rt)) {
        if (count == 0)
            return INFINI_ROCM_OK;
        IROCM_CHECK_ARG(x && y && count > 0, "all_reduce: bad buffer");
        {
            const int ord = direct_order(rt);
            if (ord != INFINI_ROCM_OK)
                return ord;
        }
        return direct_all_reduce(rt, op, dtype, x, y, count, rt->stream);
    }
    ncclDataType_t t;
    IROCM_CHECK_ARG(nccl_type(dtype, &t), "all_reduce: unsupported dtype %s", dtype_name(dtype));
    if (count == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(x && y && count > 0, "all_reduce: bad buffer");
    IROCM_NCCL(ncclAllReduce(x, y, (size_t)count, t, ops[op], (ncclComm_t)rt->comm, rt->stream));
    return INFINI_ROCM_OK;
}

// The all-reduce ordered after everything enqueued on the runtime stream so far, running on the runtime's COMM stream: work
// enqueued on the runtime stream afterwards does not wait for it until infini_rocm_comm_join. This is what lets a
// row-parallel GEMM be cut into row chunks whose all-reduces overlap the following chunks' GEMMs (bench.py tp_block, the
// plugin's MatMul -> AllReduceSum item) instead of one whole-tensor all-reduce behind the whole GEMM
// (reference: all_reduce.cc:10-33). Collectives of one communicator execute in issue order on every rank.
int infini_rocm_all_reduce_async(infiniRocmRuntime_t rt, int op, int dtype, const void *x, void *y, int64_t count) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(rt->comm || rt->dcomm, "all_reduce: communicator not initialised (call init_comm)");
    static const ncclRedOp_t ops[] = {ncclSum, ncclProd, ncclMin, ncclMax, ncclAvg};
    IROCM_CHECK_ARG(op >= 0 && op <= 4, "all_reduce: bad op %d", op);
    ncclDataType_t t;
    IROCM_CHECK_ARG(nccl_type(dtype, &t), "all_reduce: unsupported dtype %s", dtype_name(dtype));
    if (count == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(x && y && count > 0, "all_reduce: bad buffer");
    int st = ensure_comm_stream(rt);
    if (st != INFINI_ROCM_OK)
        return st;
    st = stream_fork(rt, rt->stream, rt->comm_stream);
    if (st != INFINI_ROCM_OK)
        return st;
    if (use_direct(rt)) {
        st = direct_all_reduce(rt, op, dtype, x, y, count, rt->comm_stream);
        if (st != INFINI_ROCM_OK)
            return st;
    } else {
        IROCM_NCCL(ncclAllReduce(x, y, (size_t)count, t, ops[op], (ncclComm_t)rt->comm, rt->comm_stream));
    }
    ++rt->comm_pending;
    return INFINI_ROCM_OK;
}

// The runtime stream waits for every *_async collective issued since the last join (no-op when there is none).
int infini_rocm_comm_join(infiniRocmRuntime_t rt) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    if (rt->comm_pending == 0 || !rt->comm_stream)
        return INFINI_ROCM_OK;
    rt->comm_pending = 0;
    return stream_fork(rt, rt->comm_stream, rt->stream);
}

// y[count] = sum over ranks r of x_r[rank * count .. (rank + 1) * count): x holds world_size * count elements.
// direct = 0: ncclReduceScatter (RCCL picks the algorithm). direct = 1: the one-hop exchange the fully connected xGMI mesh
// of an MI355X node allows — every rank SENDS slice j straight to rank j and receives the other ranks' slices for itself
// (grouped ncclSend / ncclRecv: all 7 links busy at once, world - 1 messages of count elements per rank instead of world - 1
// ring steps), then sums the world_size slices locally in fp32. Scratch ((world - 1) * count elements) comes from the
// runtime workspace. Sum only.
int infini_rocm_reduce_scatter(infiniRocmRuntime_t rt, int dtype, const void *x, void *y, int64_t count, int direct) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(rt->comm || rt->dcomm, "reduce_scatter: communicator not initialised (call init_comm)");
    ncclDataType_t t;
    IROCM_CHECK_ARG(nccl_type(dtype, &t), "reduce_scatter: unsupported dtype %s", dtype_name(dtype));
    if (count == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(x && y && count > 0, "reduce_scatter: bad buffer");
    if (use_direct(rt)) { // the hand-written transport IS a one-hop exchange; `direct` has nothing left to choose
        const int ord = direct_order(rt);
        if (ord != INFINI_ROCM_OK)
            return ord;
        return direct_reduce_scatter(rt, dtype, x, y, count, rt->stream);
    }
    const int world = rt->comm_world, rank = rt->comm_rank;
    if (!direct || world == 1 || !(dtype == INFINI_DT_F32 || dtype == INFINI_DT_F16 || dtype == INFINI_DT_BF16)) {
        IROCM_NCCL(ncclReduceScatter(x, y, (size_t)count, t, ncclSum, (ncclComm_t)rt->comm, rt->stream));
        return INFINI_ROCM_OK;
    }
    const size_t es = dtype_size(dtype);
    char *tmp = nullptr; // [world][count]: slot r = rank r's slice for me (slot `rank` is filled by a local copy)
    int st = infini_rocm_workspace(rt, (size_t)world * count * es, (void **)&tmp);
    if (st != INFINI_ROCM_OK)
        return st;
    IROCM_NCCL(ncclGroupStart());
    for (int r = 0; r < world; ++r) {
        if (r == rank)
            continue;
        IROCM_NCCL(ncclSend((const char *)x + (size_t)r * count * es, (size_t)count, t, r, (ncclComm_t)rt->comm, rt->stream));
        IROCM_NCCL(ncclRecv(tmp + (size_t)r * count * es, (size_t)count, t, r, (ncclComm_t)rt->comm, rt->stream));
    }
    IROCM_NCCL(ncclGroupEnd());
    IROCM_HIP(hipMemcpyAsync(tmp + (size_t)rank * count * es, (const char *)x + (size_t)rank * count * es, (size_t)count * es,
                             hipMemcpyDeviceToDevice, rt->stream));
    long g = ceil_div(count, 256);
    if (g > (long)rt->num_cu * 8) g = (long)rt->num_cu * 8;
    if (dtype == INFINI_DT_F32)
        hipLaunchKernelGGL(sum_slabs_kernel<float>, dim3((unsigned)g), dim3(256), 0, rt->stream, (const float *)tmp, (float *)y, (long)count, world);
    else if (dtype == INFINI_DT_F16)
        hipLaunchKernelGGL(sum_slabs_kernel<__half>, dim3((unsigned)g), dim3(256), 0, rt->stream, (const __half *)tmp, (__half *)y, (long)count, world);
    else
        hipLaunchKernelGGL(sum_slabs_kernel<__hip_bfloat16>, dim3((unsigned)g), dim3(256), 0, rt->stream, (const __hip_bfloat16 *)tmp,
                           (__hip_bfloat16 *)y, (long)count, world);
    IROCM_LAUNCH_CHECK("sum_slabs");
    return INFINI_ROCM_OK;
}

// y holds world_size * count elements, rank r's contribution at y + r * count.
int infini_rocm_all_gather(infiniRocmRuntime_t rt, int dtype, const void *x, void *y, int64_t count) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(rt->comm || rt->dcomm, "all_gather: communicator not initialised (call init_comm)");
    ncclDataType_t t;
    IROCM_CHECK_ARG(nccl_type(dtype, &t), "all_gather: unsupported dtype %s", dtype_name(dtype));
    if (count == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(x && y && count > 0, "all_gather: bad buffer");
    if (use_direct(rt)) {
        const int ord = direct_order(rt);
        if (ord != INFINI_ROCM_OK)
            return ord;
        return direct_all_gather(rt, x, y, (size_t)count * dtype_size(dtype), rt->stream);
    }
    IROCM_NCCL(ncclAllGather(x, y, (size_t)count, t, (ncclComm_t)rt->comm, rt->stream));
    return INFINI_ROCM_OK;
}

int infini_rocm_broadcast(infiniRocmRuntime_t rt, int dtype, const void *x, void *y, int64_t count, int root) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(rt->comm || rt->dcomm, "broadcast: communicator not initialised (call init_comm)");
    ncclDataType_t t;
    IROCM_CHECK_ARG(nccl_type(dtype, &t), "broadcast: unsupported dtype %s", dtype_name(dtype));
    IROCM_CHECK_ARG(root >= 0 && root < rt->comm_world, "broadcast: bad root %d", root);
    if (count == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(x && y && count > 0, "broadcast: bad buffer");
    if (use_direct(rt)) {
        const int ord = direct_order(rt);
        if (ord != INFINI_ROCM_OK)
            return ord;
        return direct_broadcast(rt, x, y, (size_t)count * dtype_size(dtype), root, rt->stream);
    }
    IROCM_NCCL(ncclBroadcast(x, y, (size_t)count, t, root, (ncclComm_t)rt->comm, rt->stream));
    return INFINI_ROCM_OK;
}

int infini_rocm_send(infiniRocmRuntime_t rt, int dtype, const void *x, int64_t count, int peer) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(rt->comm || rt->dcomm, "send: communicator not initialised (call init_comm)");
    ncclDataType_t t;
    IROCM_CHECK_ARG(nccl_type(dtype, &t), "send: unsupported dtype %s", dtype_name(dtype));
    IROCM_CHECK_ARG(peer >= 0 && peer < rt->comm_world && peer != rt->comm_rank, "send: bad peer %d", peer);
    if (count == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(x && count > 0, "send: bad buffer");
    if (use_direct(rt)) {
        const int ord = direct_order(rt);
        if (ord != INFINI_ROCM_OK)
            return ord;
        return direct_send(rt, x, (size_t)count * dtype_size(dtype), peer, rt->stream);
    }
    IROCM_NCCL(ncclSend(x, (size_t)count, t, peer, (ncclComm_t)rt->comm, rt->stream));
    return INFINI_ROCM_OK;
}

int infini_rocm_recv(infiniRocmRuntime_t rt, int dtype, void *y, int64_t count, int peer) {
    IROCM_CHECK_ARG(rt, "NULL runtime");
    IROCM_CHECK_ARG(rt->comm || rt->dcomm, "recv: communicator not initialised (call init_comm)");
    ncclDataType_t t;
    IROCM_CHECK_ARG(nccl_type(dtype, &t), "recv: unsupported dtype %s", dtype_name(dtype));
    IROCM_CHECK_ARG(peer >= 0 && peer < rt->comm_world && peer != rt->comm_rank, "recv: bad peer %d", peer);
    if (count == 0)
        return INFINI_ROCM_OK;
    IROCM_CHECK_ARG(y && count > 0, "recv: bad buffer");
    if (use_direct(rt)) {
        const int ord = direct_order(rt);
        if (ord != INFINI_ROCM_OK)
            return ord;
        return direct_recv(rt, y, (size_t)count * dtype_size(dtype), peer, rt->stream);
    }
    IROCM_NCCL(ncclRecv(y, (size_t)count, t, peer, (ncclComm_t)rt->comm, rt->stream));
    return INFINI_ROCM_OK;
}

} // extern "C"
