// The hand-written one-hop transport: collectives as HIP kernels over IPC-mapped peer buffers (SURVEY §5's option).
//
// Replaces, like comm.hip, NcclCommunicatorObj (reference: include/cuda/nccl_communicator.h:22-68) and the collective
// kernels AllReduceNCCL src/kernels/cuda/all_reduce.cc:8-63, AllGatherNCCL all_gather.cc:8-40, BroadcastNCCL
// broadcast.cc:8-26, SendNCCL send.cc:8-37, RecvNCCL recv.cc:8-41 — without RCCL: the 8 GPUs of an MI355X node are fully
// connected by xGMI (7 links x ~153 GB/s per GPU), so every exchange is ONE hop and a ring algorithm only wastes links.
//
// Memory. Every rank owns one uncached (fine-grained) device block, exported with hipIpcGetMemHandle and mapped by every
// peer (file rendezvous ./<name>_xgmi_<rank>.bin in the cwd, the reference's scheme). The block holds
//   ctrl   flags written by PEERS (arrival / credit words, one per (source rank, workgroup)),
//   inbox  [2 parities][world][cap]   slices pushed to me for reduction (all-reduce phase 1, reduce-scatter),
//   gbox   [2 parities][world][cap]   results / contributions pushed to me (all-reduce phase 2, all-gather),
//   bbox   [world][cap]               broadcast payload of root r,
//   pbox   [world][cap]               point-to-point payload of sender r.
// Data moves by PUSH (remote 16-byte stores, posted over xGMI; every workgroup's inner loop touches all peers so all links
// carry traffic at once); a rank only ever READS its own block.
//
// Synchronisation. A launch is G workgroups (the same G on every rank and for every call); workgroup b of rank r exchanges
// data with workgroup b of every peer ONLY (it pushes / reduces / copies sub-range b of each slice), so the whole protocol
// is per-workgroup point-to-point flags — no grid barrier, no atomics on remote memory:
//   writer: stores -> __syncthreads -> lane 0: release fence (system scope) -> s_waitcnt vmcnt(0) -> flag[me][b] = s in the
//           peer's ctrl;      reader: lane 0 polls its OWN ctrl until flag[src][b] >= s (bounded: a time limit sets the
//           communicator's error word instead of hanging the GPU) -> acquire fence -> __syncthreads -> loads.
// s is a per-workgroup call counter kept in device memory (incremented by the kernel itself), so a launch carries no
// sequence number and a hipGraph that captured it can be replayed.
// Buffer reuse. all-reduce / all-gather / reduce-scatter make every rank wait for every other rank's push of the same call,
// and a rank issues call s + 1's pushes only after it finished call s; with the boxes double-buffered by the parity of s a
// rank can therefore never overwrite data a slower peer still reads (it cannot be two calls ahead). Broadcast and
// send / recv are one-sided: they use their own boxes with CREDITS (the receiver acknowledges into the sender's ctrl; the
// sender waits for the previous message's acknowledgement before it pushes the next one).
//
// Ranks may share a device (every rank opens device `rt->device`): RCCL refuses that, this transport does not — which is
// how the reference's multi-rank collective tests (test_cuda_all_reduce.cc:38-106, ...) run on a one-GPU box.
#include "common.h"

#include <chrono>
#include <fstream>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

namespace irocm {

constexpr int kDMaxRanks = 8;
constexpr int kDGrid = 32;      // workgroups per launch (fixed: part of the protocol)
constexpr int kDThreads = 512;

struct DirectCtrl { // written by peers (and read by me); lives at the start of my exported block
    unsigned flagS[kDMaxRanks][kDGrid]; // inbox arrivals
    unsigned flagG[kDMaxRanks][kDGrid]; // gbox arrivals
    unsigned flagB[kDMaxRanks][kDGrid]; // bbox arrivals, indexed by root
    unsigned ackB[kDMaxRanks][kDGrid];  // broadcast credits, indexed by the acknowledging peer (in the ROOT's ctrl)
    unsigned flagP[kDMaxRanks][kDGrid]; // pbox arrivals, indexed by sender
    unsigned ackP[kDMaxRanks][kDGrid];  // p2p credits, indexed by the acknowledging receiver (in the SENDER's ctrl)
    unsigned error;                     // set by my own kernels on a time limit
};

struct DirectLocal { // private device memory: the per-workgroup call counters
    unsigned seq[kDGrid];                  // all-reduce / all-gather / reduce-scatter calls
    unsigned bseq[kDGrid];                 // broadcast calls
    unsigned broot_last[kDGrid];           // bseq of my last broadcast as root
    unsigned sseq[kDMaxRanks][kDGrid];     // messages sent to peer
    unsigned rseq[kDMaxRanks][kDGrid];     // messages received from peer
};

struct DirectArgs { // by value into every kernel
    char *base[kDMaxRanks]; // the exported block of rank r as mapped in THIS process (own block for r == rank)
    DirectLocal *local;
    long long timeout_ticks; // wall_clock64 ticks (100 MHz)
    size_t cap;              // bytes per box slot
    int world, rank;
};

constexpr size_t kCtrlBytes = (sizeof(DirectCtrl) + 4095) & ~(size_t)4095;

int direct_check(infiniRocmRuntime *rt);
int direct_all_gather(infiniRocmRuntime *rt, const void *x, void *y, size_t bytes, hipStream_t st);

struct DirectComm {
    DirectArgs args;
    void *block = nullptr;
    size_t block_bytes = 0;
    void *opened[kDMaxRanks] = {};
    std::string my_file;
};

__device__ __forceinline__ DirectCtrl *ctrl_of(const DirectArgs &a, int r) { return (DirectCtrl *)a.base[r]; }
__device__ __forceinline__ char *inbox_of(const DirectArgs &a, int r, int par, int src) {
    return a.base[r] + kCtrlBytes + ((size_t)par * a.world + src) * a.cap;
}
__device__ __forceinline__ char *gbox_of(const DirectArgs &a, int r, int par, int src) {
    return a.base[r] + kCtrlBytes + ((size_t)(2 + par) * a.world + src) * a.cap;
}
__device__ __forceinline__ char *bbox_of(const DirectArgs &a, int r, int root) {
    return a.base[r] + kCtrlBytes + ((size_t)4 * a.world + root) * a.cap;
}
__device__ __forceinline__ char *pbox_of(const DirectArgs &a, int r, int src) {
    return a.base[r] + kCtrlBytes + ((size_t)5 * a.world + src) * a.cap;
}

// lane 0 only. Returns after *f >= want (flags grow monotonically) or after the time limit (error word set).
__device__ __forceinline__ void wait_flag(const DirectArgs &a, unsigned *f, unsigned want) {
    // a communicator that already ran into its time limit does not wait again: every later wait of this rank returns at once,
    // so a peer that never shows up costs ONE time limit per rank, not one per flag and call
    if (__hip_atomic_load(&ctrl_of(a, a.rank)->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)
        return;
    const long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - want) < 0) {
        __builtin_amdgcn_s_sleep(4);
        if (wall_clock64() - t0 > a.timeout_ticks) {
            __hip_atomic_store(&ctrl_of(a, a.rank)->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
}
// lane 0 only, after a __syncthreads() that follows the workgroup's remote stores
__device__ __forceinline__ void publish(unsigned *remote_flag, unsigned s) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (hipcc may drop the fence's own wait: MI355X_MICROARCH, compiler hazard)
    __hip_atomic_store(remote_flag, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); }

// sub-range b of `units` units
__device__ __forceinline__ void sub_range(long units, int b, long &u0, long &u1) {
    u0 = units * b / kDGrid;
    u1 = units * (b + 1) / kDGrid;
}

template <typename T, int VEC> struct alignas(sizeof(T) * VEC) Pack { T v[VEC]; };

// copy units [u0, u1) of a `len`-element range (unit = VEC elements; the last unit may be partial)
template <typename T, int VEC>
__device__ __forceinline__ void copy_units(const T *src, T *dst, long len, long u0, long u1) {
    for (long u = u0 + threadIdx.x; u < u1; u += kDThreads) {
        const long e = u * VEC;
        if (VEC > 1 && e + VEC <= len) {
            *reinterpret_cast<Pack<T, VEC> *>(dst + e) = *reinterpret_cast<const Pack<T, VEC> *>(src + e);
        } else {
            for (long i = e; i < len && i < e + VEC; ++i)
                dst[i] = src[i];
        }
    }
}

template <typename T> struct Acc { using type = long long; };
template <> struct Acc<float> { using type = float; };
template <> struct Acc<__half> { using type = float; };
template <> struct Acc<__hip_bfloat16> { using type = float; };
template <> struct Acc<double> { using type = double; };
template <> struct Acc<unsigned long long> { using type = unsigned long long; };

template <typename T> __device__ __forceinline__ typename Acc<T>::type to_acc(T v) { return (typename Acc<T>::type)v; }
template <> __device__ __forceinline__ float to_acc<__half>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ float to_acc<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }
template <typename T> __device__ __forceinline__ T from_acc(typename Acc<T>::type v) { return (T)v; }
template <> __device__ __forceinline__ __half from_acc<__half>(float v) { return __float2half(v); }
template <> __device__ __forceinline__ __hip_bfloat16 from_acc<__hip_bfloat16>(float v) { return __float2bfloat16(v); }

template <typename A> __device__ __forceinline__ A red(int op, A a, A b) {
    switch (op) {
    case 1: return a * b;
    case 2: return b < a ? b : a;
    case 3: return b > a ? b : a;
    default: return a + b; // sum, avg
    }
}

// All-reduce (mode 0) / reduce-scatter (mode 1: y receives my reduced slice only).
// Slice j starts at x + j * slice and holds fixed_len elements when fixed_len >= 0 (reduce-scatter: the caller's layout),
// else min(slice, count - j * slice) (all-reduce: the launcher cuts `count` into slices that are multiples of VEC).
// Values are combined in rank order 0 .. world - 1 in the accumulator type (fp32 for 16-bit floats, exact 64-bit integers
// for the integer types), rounded once; rank j computes slice j and pushes the result to everybody, so every rank ends
// with the same bits.
template <typename T, int VEC>
__global__ __launch_bounds__(kDThreads) void direct_reduce_kernel(DirectArgs a, const T *x, T *y, long count, long slice, long fixed_len,
                                                                  int op, int mode) {
    using A = typename Acc<T>::type;
    const int b = blockIdx.x, me = a.rank, n = a.world;
    __shared__ unsigned s_sh;
    if (threadIdx.x == 0)
        s_sh = a.local->seq[b] + 1;
    __syncthreads();
    const unsigned s = s_sh;
    const int par = s & 1;
    auto slice_len = [&](int j) {
        if (fixed_len >= 0)
            return fixed_len;
        const long rest = count - (long)j * slice;
        return rest < 0 ? 0 : (rest < slice ? rest : slice);
    };
    // ---- phase 1: push slice j of my x into inbox[par][me] of rank j (all peers inside the unit loop: every link busy)
    {
        long maxu = 0;
        for (int j = 0; j < n; ++j)
            if (j != me) {
                const long units = (slice_len(j) + VEC - 1) / VEC;
                long u0, u1;
                sub_range(units, b, u0, u1);
                maxu = u1 - u0 > maxu ? u1 - u0 : maxu;
            }
        for (long i = threadIdx.x; i < maxu; i += kDThreads)
            for (int d = 1; d < n; ++d) {
                const int j = (me + d) % n;
                const long len = slice_len(j), units = (len + VEC - 1) / VEC;
                long u0, u1;
                sub_range(units, b, u0, u1);
                const long u = u0 + i;
                if (u >= u1)
                    continue;
                const T *src = x + (long)j * slice;
                T *dst = (T *)inbox_of(a, j, par, me);
                const long e = u * VEC;
                if (VEC > 1 && e + VEC <= len) {
                    *reinterpret_cast<Pack<T, VEC> *>(dst + e) = *reinterpret_cast<const Pack<T, VEC> *>(src + e);
                } else {
                    for (long q = e; q < len && q < e + VEC; ++q)
                        dst[q] = src[q];
                }
            }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int d = 1; d < n; ++d)
            publish(&ctrl_of(a, (me + d) % n)->flagS[me][b], s);
        for (int d = 1; d < n; ++d)
            wait_flag(a, &ctrl_of(a, me)->flagS[(me + d) % n][b], s);
        acquire();
    }
    __syncthreads();
    // ---- phase 2: reduce my slice; mode 0: result to my y and into gbox[par][me] of every peer
    {
        const long len = slice_len(me), units = (len + VEC - 1) / VEC;
        long u0, u1;
        sub_range(units, b, u0, u1);
        const T *mine = x + (long)me * slice;
        T *out = mode == 0 ? y + (long)me * slice : y;
        for (long u = u0 + threadIdx.x; u < u1; u += kDThreads) {
            const long e = u * VEC;
            const int lim = (int)(len - e < VEC ? len - e : VEC);
            A acc[VEC];
            for (int r = 0; r < n; ++r) {
                const T *src = r == me ? mine : (const T *)inbox_of(a, me, par, r);
                T v[VEC];
                if (VEC > 1 && lim == VEC) {
                    *reinterpret_cast<Pack<T, VEC> *>(v) = *reinterpret_cast<const Pack<T, VEC> *>(src + e);
                } else {
                    for (int q = 0; q < lim; ++q)
                        v[q] = src[e + q];
                }
#pragma unroll
                for (int q = 0; q < VEC; ++q)
                    if (q < lim)
                        acc[q] = r == 0 ? to_acc<T>(v[q]) : red<A>(op, acc[q], to_acc<T>(v[q]));
            }
            T res[VEC];
#pragma unroll
            for (int q = 0; q < VEC; ++q)
                if (q < lim)
                    res[q] = from_acc<T>(op == 4 ? (A)(acc[q] / (A)n) : acc[q]);
            if (VEC > 1 && lim == VEC) {
                *reinterpret_cast<Pack<T, VEC> *>(out + e) = *reinterpret_cast<const Pack<T, VEC> *>(res);
                if (mode == 0)
                    for (int d = 1; d < n; ++d)
                        *reinterpret_cast<Pack<T, VEC> *>((T *)gbox_of(a, (me + d) % n, par, me) + e) = *reinterpret_cast<const Pack<T, VEC> *>(res);
            } else {
                for (int q = 0; q < lim; ++q) {
                    out[e + q] = res[q];
                    if (mode == 0)
                        for (int d = 1; d < n; ++d)
                            ((T *)gbox_of(a, (me + d) % n, par, me))[e + q] = res[q];
                }
            }
        }
    }
    if (mode == 0) {
        __syncthreads();
        if (threadIdx.x == 0)
            for (int d = 1; d < n; ++d)
                publish(&ctrl_of(a, (me + d) % n)->flagG[me][b], s);
        // ---- phase 3: the other ranks' result slices from my gbox into y
        for (int d = 1; d < n; ++d) {
            const int src = (me + d) % n;
            if (threadIdx.x == 0) {
                wait_flag(a, &ctrl_of(a, me)->flagG[src][b], s);
                acquire();
            }
            __syncthreads();
            const long len = slice_len(src), units = (len + VEC - 1) / VEC;
            long u0, u1;
            sub_range(units, b, u0, u1);
            copy_units<T, VEC>((const T *)gbox_of(a, me, par, src), y + (long)src * slice, len, u0, u1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0)
        a.local->seq[b] = s;
}

// All-gather of `len` elements per rank: y + r * ystride receives rank r's elements (bytes as 16-byte / 1-byte units).
template <int VEC>
__global__ __launch_bounds__(kDThreads) void direct_gather_kernel(DirectArgs a, const char *x, char *y, long len, long ystride) {
    using T = char;
    const int b = blockIdx.x, me = a.rank, n = a.world;
    __shared__ unsigned s_sh;
    if (threadIdx.x == 0)
        s_sh = a.local->seq[b] + 1;
    __syncthreads();
    const unsigned s = s_sh;
    const int par = s & 1;
    const long units = (len + VEC - 1) / VEC;
    long u0, u1;
    sub_range(units, b, u0, u1);
    for (long u = u0 + threadIdx.x; u < u1; u += kDThreads) {
        const long e = u * VEC;
        if (VEC > 1 && e + VEC <= len) {
            const Pack<T, VEC> v = *reinterpret_cast<const Pack<T, VEC> *>(x + e);
            for (int d = 1; d < n; ++d)
                *reinterpret_cast<Pack<T, VEC> *>(gbox_of(a, (me + d) % n, par, me) + e) = v;
            *reinterpret_cast<Pack<T, VEC> *>(y + (long)me * ystride + e) = v;
        } else {
            for (long q = e; q < len && q < e + VEC; ++q) {
                const char v = x[q];
                for (int d = 1; d < n; ++d)
                    gbox_of(a, (me + d) % n, par, me)[q] = v;
                y[(long)me * ystride + q] = v;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int d = 1; d < n; ++d)
            publish(&ctrl_of(a, (me + d) % n)->flagG[me][b], s);
    for (int d = 1; d < n; ++d) {
        const int src = (me + d) % n;
        if (threadIdx.x == 0) {
            wait_flag(a, &ctrl_of(a, me)->flagG[src][b], s);
            acquire();
        }
        __syncthreads();
        copy_units<T, VEC>(gbox_of(a, me, par, src), y + (long)src * ystride, len, u0, u1);
    }
    __syncthreads();
    if (threadIdx.x == 0)
        a.local->seq[b] = s;
}

template <int VEC>
__global__ __launch_bounds__(kDThreads) void direct_broadcast_kernel(DirectArgs a, const char *x, char *y, long len, int root) {
    using T = char;
    const int b = blockIdx.x, me = a.rank, n = a.world;
    __shared__ unsigned s_sh;
    if (threadIdx.x == 0)
        s_sh = a.local->bseq[b] + 1;
    __syncthreads();
    const unsigned s = s_sh;
    const long units = (len + VEC - 1) / VEC;
    long u0, u1;
    sub_range(units, b, u0, u1);
    if (me == root) {
        if (threadIdx.x == 0) { // credit: my previous broadcast has been copied out of every peer's bbox[me]
            const unsigned last = a.local->broot_last[b];
            for (int d = 1; d < n; ++d)
                wait_flag(a, &ctrl_of(a, me)->ackB[(me + d) % n][b], last);
        }
        __syncthreads();
        for (long u = u0 + threadIdx.x; u < u1; u += kDThreads) {
            const long e = u * VEC;
            if (VEC > 1 && e + VEC <= len) {
                const Pack<T, VEC> v = *reinterpret_cast<const Pack<T, VEC> *>(x + e);
                for (int d = 1; d < n; ++d)
                    *reinterpret_cast<Pack<T, VEC> *>(bbox_of(a, (me + d) % n, root) + e) = v;
                if (y != x)
                    *reinterpret_cast<Pack<T, VEC> *>(y + e) = v;
            } else {
                for (long q = e; q < len && q < e + VEC; ++q) {
                    const char v = x[q];
                    for (int d = 1; d < n; ++d)
                        bbox_of(a, (me + d) % n, root)[q] = v;
                    y[q] = v;
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int d = 1; d < n; ++d)
                publish(&ctrl_of(a, (me + d) % n)->flagB[root][b], s);
            a.local->broot_last[b] = s;
        }
    } else {
        if (threadIdx.x == 0) {
            wait_flag(a, &ctrl_of(a, me)->flagB[root][b], s);
            acquire();
        }
        __syncthreads();
        copy_units<T, VEC>(bbox_of(a, me, root), y, len, u0, u1);
        __syncthreads(); // (carries the wait for this workgroup's loads: the values are in registers / stored)
        if (threadIdx.x == 0)
            publish(&ctrl_of(a, root)->ackB[me][b], s);
    }
    if (threadIdx.x == 0)
        a.local->bseq[b] = s;
}

template <int VEC>
__global__ __launch_bounds__(kDThreads) void direct_send_kernel(DirectArgs a, const char *x, long len, int peer) {
    using T = char;
    const int b = blockIdx.x, me = a.rank;
    __shared__ unsigned s_sh;
    if (threadIdx.x == 0) {
        s_sh = a.local->sseq[peer][b] + 1;
        wait_flag(a, &ctrl_of(a, me)->ackP[peer][b], s_sh - 1); // credit: the previous message left the peer's pbox[me]
    }
    __syncthreads();
    const unsigned s = s_sh;
    const long units = (len + VEC - 1) / VEC;
    long u0, u1;
    sub_range(units, b, u0, u1);
    copy_units<T, VEC>(x, pbox_of(a, peer, me), len, u0, u1);
    __syncthreads();
    if (threadIdx.x == 0) {
        publish(&ctrl_of(a, peer)->flagP[me][b], s);
        a.local->sseq[peer][b] = s;
    }
}

template <int VEC>
__global__ __launch_bounds__(kDThreads) void direct_recv_kernel(DirectArgs a, char *y, long len, int peer) {
    using T = char;
    const int b = blockIdx.x, me = a.rank;
    __shared__ unsigned s_sh;
    if (threadIdx.x == 0) {
        s_sh = a.local->rseq[peer][b] + 1;
        wait_flag(a, &ctrl_of(a, me)->flagP[peer][b], s_sh);
        acquire();
    }
    __syncthreads();
    const unsigned s = s_sh;
    const long units = (len + VEC - 1) / VEC;
    long u0, u1;
    sub_range(units, b, u0, u1);
    copy_units<T, VEC>(pbox_of(a, me, peer), y, len, u0, u1);
    __syncthreads();
    if (threadIdx.x == 0) {
        publish(&ctrl_of(a, peer)->ackP[me][b], s);
        a.local->rseq[peer][b] = s;
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
struct DirectHello {
    char magic[8];
    int world, rank, grid, pid;
    unsigned long long cap;
    hipIpcMemHandle_t handle;
};

static bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

int direct_init(infiniRocmRuntime *rt, const char *name, int world, int rank) {
    IROCM_CHECK_ARG(world >= 1 && world <= kDMaxRanks, "direct transport: world size %d (1..%d: one xGMI node)", world, kDMaxRanks);
    IROCM_CHECK_ARG(rt->dcomm == nullptr, "direct communicator already initialised");
    IROCM_HIP(hipSetDevice(rt->device));
    auto *dc = new DirectComm();
    const char *cap_env = std::getenv("INFINI_ROCM_DIRECT_CAP_MB");
    const char *to_env = std::getenv("INFINI_ROCM_DIRECT_TIMEOUT_S");
    size_t cap = (size_t)(cap_env ? std::max(1, std::atoi(cap_env)) : 8) << 20;
    const double timeout_s = to_env ? std::atof(to_env) : 20.0;
    dc->block_bytes = kCtrlBytes + (size_t)6 * world * cap;
    // uncached / fine-grained: a peer's stores must not be shadowed by stale lines in my L2 (RCCL allocates its buffers the
    // same way); plain hipMalloc only as a last resort (enough when all ranks share one device).
    hipError_t e = hipExtMallocWithFlags(&dc->block, dc->block_bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(&dc->block, dc->block_bytes, hipDeviceMallocFinegrained);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipMalloc(&dc->block, dc->block_bytes);
    }
    if (e != hipSuccess) {
        const size_t want = dc->block_bytes;
        delete dc;
        IROCM_FAIL(INFINI_ROCM_OUT_OF_MEMORY, "direct transport: cannot allocate %zu bytes: %s", want, hipGetErrorString(e));
    }
    auto fail = [&](int code, const std::string &msg) {
        for (int r = 0; r < kDMaxRanks; ++r)
            if (dc->opened[r])
                (void)hipIpcCloseMemHandle(dc->opened[r]);
        if (dc->args.local)
            (void)hipFree(dc->args.local);
        (void)hipFree(dc->block);
        if (!dc->my_file.empty())
            (void)unlink(dc->my_file.c_str());
        delete dc;
        set_error("%s", msg.c_str());
        return code;
    };
    if (hipMemset(dc->block, 0, kCtrlBytes) != hipSuccess || hipMalloc((void **)&dc->args.local, sizeof(DirectLocal)) != hipSuccess ||
        hipMemset(dc->args.local, 0, sizeof(DirectLocal)) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
        return fail(INFINI_ROCM_HIP_ERROR, "direct transport: cannot initialise the control block");
    dc->args.world = world;
    dc->args.rank = rank;
    dc->args.cap = cap;
    dc->args.timeout_ticks = (long long)(timeout_s * 1e8);
    dc->args.base[rank] = (char *)dc->block;
    if (world > 1) {
        DirectHello me{};
        memcpy(me.magic, "IROCMXG1", 8);
        me.world = world, me.rank = rank, me.grid = kDGrid, me.pid = (int)getpid(), me.cap = cap;
        if ((e = hipIpcGetMemHandle(&me.handle, dc->block)) != hipSuccess)
            return fail(INFINI_ROCM_HIP_ERROR, std::string("direct transport: hipIpcGetMemHandle failed: ") + hipGetErrorString(e) +
                                                   " (is HSA_ENABLE_IPC_MODE_LEGACY=0 exported?)");
        auto path = [&](int r) { return std::string("./") + name + "_xgmi_" + std::to_string(r) + ".bin"; };
        dc->my_file = path(rank);
        {
            const std::string tmp = dc->my_file + ".tmp";
            std::ofstream ofs(tmp, std::ios::binary | std::ios::trunc);
            ofs.write((const char *)&me, sizeof(me));
            ofs.close();
            if (rename(tmp.c_str(), dc->my_file.c_str()) != 0)
                return fail(INFINI_ROCM_RCCL_ERROR, "direct transport: cannot publish " + dc->my_file);
        }
        const auto begin = std::chrono::steady_clock::now();
        for (int r = 0; r < world; ++r) {
            if (r == rank)
                continue;
            struct stat st;
            while (stat(path(r).c_str(), &st) != 0 || (size_t)st.st_size < sizeof(DirectHello)) {
                if (std::chrono::steady_clock::now() > begin + std::chrono::seconds(120))
                    return fail(INFINI_ROCM_RCCL_ERROR, "direct transport: time limit (120 s) exceeded waiting for " + path(r));
                std::this_thread::sleep_for(std::chrono::milliseconds(50));
            }
            DirectHello peer{};
            std::ifstream ifs(path(r), std::ios::binary);
            ifs.read((char *)&peer, sizeof(peer));
            if (memcmp(peer.magic, "IROCMXG1", 8) || peer.world != world || peer.rank != r || peer.grid != kDGrid || peer.cap != cap)
                return fail(INFINI_ROCM_RCCL_ERROR, "direct transport: " + path(r) + " does not match this job (stale file or different settings)");
            if ((e = hipIpcOpenMemHandle(&dc->opened[r], peer.handle, hipIpcMemLazyEnablePeerAccess)) != hipSuccess)
                return fail(INFINI_ROCM_HIP_ERROR, std::string("direct transport: hipIpcOpenMemHandle(rank ") + std::to_string(r) + ") failed: " +
                                                       hipGetErrorString(e));
            dc->args.base[r] = (char *)dc->opened[r];
        }
    }
    rt->dcomm = dc;
    rt->comm_world = world;
    rt->comm_rank = rank;
    if (world > 1) {
        // handshake: a one-word all-gather proves every mapping before anybody removes its rendezvous file
        unsigned *w = nullptr;
        int st = INFINI_ROCM_OK;
        if (hipMalloc((void **)&w, sizeof(unsigned) * (world + 1)) != hipSuccess)
            st = INFINI_ROCM_OUT_OF_MEMORY;
        if (st == INFINI_ROCM_OK)
            st = direct_all_gather(rt, w, w + 1, sizeof(unsigned), rt->stream);
        if (st == INFINI_ROCM_OK && hipStreamSynchronize(rt->stream) != hipSuccess)
            st = INFINI_ROCM_HIP_ERROR;
        if (st == INFINI_ROCM_OK)
            st = direct_check(rt);
        if (w)
            (void)hipFree(w);
        (void)unlink(dc->my_file.c_str());
        dc->my_file.clear();
        if (st != INFINI_ROCM_OK) {
            const std::string msg = "direct transport: the handshake with the peers failed";
            rt->dcomm = nullptr;
            rt->comm_world = 1, rt->comm_rank = 0;
            return fail(st, msg);
        }
    }
    return INFINI_ROCM_OK;
}

int direct_destroy(infiniRocmRuntime *rt) {
    auto *dc = (DirectComm *)rt->dcomm;
    if (!dc)
        return INFINI_ROCM_OK;
    (void)hipSetDevice(rt->device);
    (void)hipStreamSynchronize(rt->stream);
    if (rt->comm_stream)
        (void)hipStreamSynchronize(rt->comm_stream);
    for (int r = 0; r < kDMaxRanks; ++r)
        if (dc->opened[r])
            (void)hipIpcCloseMemHandle(dc->opened[r]);
    (void)hipFree(dc->args.local);
    (void)hipFree(dc->block);
    if (!dc->my_file.empty())
        (void)unlink(dc->my_file.c_str());
    delete dc;
    rt->dcomm = nullptr;
    return INFINI_ROCM_OK;
}

// blocking: reads the error word my kernels set when a peer did not show up within the time limit
int direct_check(infiniRocmRuntime *rt) {
    auto *dc = (DirectComm *)rt->dcomm;
    if (!dc)
        return INFINI_ROCM_OK;
    unsigned err = 0;
    IROCM_HIP(hipMemcpy(&err, &((DirectCtrl *)dc->block)->error, sizeof(err), hipMemcpyDeviceToHost));
    if (err)
        IROCM_FAIL(INFINI_ROCM_RCCL_ERROR, "direct transport: a peer did not arrive within the time limit (results of the affected "
                                           "collectives are undefined)");
    return INFINI_ROCM_OK;
}

template <typename T>
static int launch_reduce(DirectComm *dc, const void *x, void *y, long count, long slice, long fixed_len, int op, int mode, hipStream_t st) {
    constexpr int V = 16 / (int)sizeof(T);
    if (aligned16(x) && aligned16(y) && (slice * sizeof(T)) % 16 == 0)
        hipLaunchKernelGGL((direct_reduce_kernel<T, V>), dim3(kDGrid), dim3(kDThreads), 0, st, dc->args, (const T *)x, (T *)y, count, slice,
                           fixed_len, op, mode);
    else
        hipLaunchKernelGGL((direct_reduce_kernel<T, 1>), dim3(kDGrid), dim3(kDThreads), 0, st, dc->args, (const T *)x, (T *)y, count, slice,
                           fixed_len, op, mode);
    IROCM_LAUNCH_CHECK("direct_reduce");
    return INFINI_ROCM_OK;
}

static int reduce_typed(DirectComm *dc, int dtype, const void *x, void *y, long count, long slice, long fixed_len, int op, int mode,
                        hipStream_t st) {
    switch (dtype) {
    case INFINI_DT_F32: return launch_reduce<float>(dc, x, y, count, slice, fixed_len, op, mode, st);
    case INFINI_DT_F16: return launch_reduce<__half>(dc, x, y, count, slice, fixed_len, op, mode, st);
    case INFINI_DT_BF16: return launch_reduce<__hip_bfloat16>(dc, x, y, count, slice, fixed_len, op, mode, st);
    case INFINI_DT_F64: return launch_reduce<double>(dc, x, y, count, slice, fixed_len, op, mode, st);
    case INFINI_DT_I8: return launch_reduce<signed char>(dc, x, y, count, slice, fixed_len, op, mode, st);
    case INFINI_DT_U8: case INFINI_DT_BOOL: return launch_reduce<unsigned char>(dc, x, y, count, slice, fixed_len, op, mode, st);
    case INFINI_DT_I32: return launch_reduce<int>(dc, x, y, count, slice, fixed_len, op, mode, st);
    case INFINI_DT_U32: return launch_reduce<unsigned>(dc, x, y, count, slice, fixed_len, op, mode, st);
    case INFINI_DT_I64: return launch_reduce<long long>(dc, x, y, count, slice, fixed_len, op, mode, st);
    case INFINI_DT_U64: return launch_reduce<unsigned long long>(dc, x, y, count, slice, fixed_len, op, mode, st);
    default: IROCM_FAIL(INFINI_ROCM_UNSUPPORTED, "direct transport: unsupported dtype %s", dtype_name(dtype));
    }
}

int direct_all_reduce(infiniRocmRuntime *rt, int op, int dtype, const void *x, void *y, int64_t count, hipStream_t st) {
    auto *dc = (DirectComm *)rt->dcomm;
    const size_t es = dtype_size(dtype);
    IROCM_CHECK_ARG(es, "direct transport: unsupported dtype %s", dtype_name(dtype));
    const int n = dc->args.world;
    const long vec = 16 / (long)es, cap_elems = (long)(dc->args.cap / es);
    // pieces of at most world * cap: each piece is one protocol call with slices of <= cap
    for (int64_t off = 0; off < count;) {
        const int64_t piece = std::min<int64_t>(count - off, (int64_t)n * cap_elems);
        long slice = (long)((piece + n - 1) / n);
        slice = (slice + vec - 1) / vec * vec;
        const int rc = reduce_typed(dc, dtype, (const char *)x + off * es, (char *)y + off * es, (long)piece, slice, -1, op, 0, st);
        if (rc != INFINI_ROCM_OK)
            return rc;
        off += piece;
    }
    return INFINI_ROCM_OK;
}

// x: world slices of `count` elements, y: my reduced slice (sum)
int direct_reduce_scatter(infiniRocmRuntime *rt, int dtype, const void *x, void *y, int64_t count, hipStream_t st) {
    auto *dc = (DirectComm *)rt->dcomm;
    const size_t es = dtype_size(dtype);
    IROCM_CHECK_ARG(es, "direct transport: unsupported dtype %s", dtype_name(dtype));
    const long vec = 16 / (long)es, cap_elems = (long)(dc->args.cap / es) / vec * vec;
    // a piece = the same sub-range [off, off + len) of every rank's slice (slice stride = the caller's `count`)
    for (int64_t off = 0; off < count;) {
        const int64_t len = std::min<int64_t>(count - off, cap_elems);
        const int rc = reduce_typed(dc, dtype, (const char *)x + off * es, (char *)y + off * es, 0, (long)count, (long)len, 0, 1, st);
        if (rc != INFINI_ROCM_OK)
            return rc;
        off += len;
    }
    return INFINI_ROCM_OK;
}

// bytes-level all-gather: y + r * bytes receives rank r's `bytes`
int direct_all_gather(infiniRocmRuntime *rt, const void *x, void *y, size_t bytes, hipStream_t st) {
    auto *dc = (DirectComm *)rt->dcomm;
    for (size_t off = 0; off < bytes;) {
        const size_t len = std::min(bytes - off, dc->args.cap);
        const char *xs = (const char *)x + off;
        char *ys = (char *)y + off;
        const bool al = aligned16(xs) && aligned16(ys) && bytes % 16 == 0;
        if (al)
            hipLaunchKernelGGL((direct_gather_kernel<16>), dim3(kDGrid), dim3(kDThreads), 0, st, dc->args, xs, ys, (long)len, (long)bytes);
        else
            hipLaunchKernelGGL((direct_gather_kernel<1>), dim3(kDGrid), dim3(kDThreads), 0, st, dc->args, xs, ys, (long)len, (long)bytes);
        IROCM_LAUNCH_CHECK("direct_gather");
        off += len;
    }
    return INFINI_ROCM_OK;
}

int direct_broadcast(infiniRocmRuntime *rt, const void *x, void *y, size_t bytes, int root, hipStream_t st) {
    auto *dc = (DirectComm *)rt->dcomm;
    for (size_t off = 0; off < bytes;) {
        const size_t len = std::min(bytes - off, dc->args.cap);
        const char *xs = (const char *)x + off;
        char *ys = (char *)y + off;
        if (aligned16(xs) && aligned16(ys))
            hipLaunchKernelGGL((direct_broadcast_kernel<16>), dim3(kDGrid), dim3(kDThreads), 0, st, dc->args, xs, ys, (long)len, root);
        else
            hipLaunchKernelGGL((direct_broadcast_kernel<1>), dim3(kDGrid), dim3(kDThreads), 0, st, dc->args, xs, ys, (long)len, root);
        IROCM_LAUNCH_CHECK("direct_broadcast");
        off += len;
    }
    return INFINI_ROCM_OK;
}

int direct_send(infiniRocmRuntime *rt, const void *x, size_t bytes, int peer, hipStream_t st) {
    auto *dc = (DirectComm *)rt->dcomm;
    for (size_t off = 0; off < bytes;) {
        const size_t len = std::min(bytes - off, dc->args.cap);
        const char *xs = (const char *)x + off;
        if (aligned16(xs))
            hipLaunchKernelGGL((direct_send_kernel<16>), dim3(kDGrid), dim3(kDThreads), 0, st, dc->args, xs, (long)len, peer);
        else
            hipLaunchKernelGGL((direct_send_kernel<1>), dim3(kDGrid), dim3(kDThreads), 0, st, dc->args, xs, (long)len, peer);
        IROCM_LAUNCH_CHECK("direct_send");
        off += len;
    }
    return INFINI_ROCM_OK;
}

int direct_recv(infiniRocmRuntime *rt, void *y, size_t bytes, int peer, hipStream_t st) {
    auto *dc = (DirectComm *)rt->dcomm;
    for (size_t off = 0; off < bytes;) {
        const size_t len = std::min(bytes - off, dc->args.cap);
        char *ys = (char *)y + off;
        if (aligned16(ys))
            hipLaunchKernelGGL((direct_recv_kernel<16>), dim3(kDGrid), dim3(kDThreads), 0, st, dc->args, ys, (long)len, peer);
        else
            hipLaunchKernelGGL((direct_recv_kernel<1>), dim3(kDGrid), dim3(kDThreads), 0, st, dc->args, ys, (long)len, peer);
        IROCM_LAUNCH_CHECK("direct_recv");
        off += len;
    }
    return INFINI_ROCM_OK;
}

} // namespace irocm
