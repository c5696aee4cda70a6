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
//   writer: stores -> EVERY thread: s_waitcnt vmcnt(0) (release_stores: its stores have left the wave) -> __syncthreads ->
//           lane 0: release fence (system scope) -> s_waitcnt vmcnt(0) -> flag[me][b] = s in the peer's ctrl;      reader: lane 0 polls its OWN ctrl until flag[src][b] >= s (bounded: a time limit sets the
//           communicator's error word instead of hanging the GPU) -> acquire fence -> __syncthreads -> loads.
// s is a per-workgroup call counter kept in device memory (incremented by the kernel itself), so a launch carries no
// sequence number and a hipGraph that captured it can be replayed.
// Buffer reuse. all-reduce / all-gather / reduce-scatter make every rank wait for every other rank's push of the same call,
// and a rank issues call s + 1's pushes only after it finished call s; the boxes are double-buffered by the parity of s. The
// per-workgroup flags alone do NOT make that safe: workgroup b of rank A may be at call s + 1 while workgroup b' of rank B is
// still reading call s - 1's boxes (same parity), and when the two calls differ in length their ranges differ, so A's range b can
// overlap B's range b'. Every kernel of the family therefore starts by waiting until EVERY workgroup of every peer has published
// call s - 1 (wait_all_workgroups on flagS / flagG in its own ctrl — a peer's workgroup publishes call s - 1 only after it left
// call s - 2). Broadcast and send / recv are one-sided: they use their own boxes with CREDITS (the receiver's workgroups
// acknowledge into the sender's ctrl; before it pushes the next message the sender waits for the acknowledgement of the
// previous one from ALL of the receiver's workgroups, for the same reason). The range of workgroup b is cut in 16-byte units
// whatever vector width a rank's kernel uses (elem_range), so writer and reader always agree on it.
//
// Ranks may share a device (every rank opens device `rt->device`): RCCL refuses that, this transport does not — which is
// how the reference's multi-rank collective tests (test_cuda_all_reduce.cc:38-106, ...) run on a one-GPU box.
#include "common.h"

#include <chrono>
#include <fstream>
#include <signal.h>
#include <cerrno>
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
// EVERY thread, in front of the __syncthreads() that precedes publish(): its own remote stores are performed at system scope.
// The barrier does not do that — for a workgroup-scope barrier hipcc emits no vmcnt wait (outside tgsplit mode the waves of a
// workgroup share their CU's L1, which is all workgroup scope asks for) — and the fence inside publish() covers the stores of lane
// 0's wave only: without this the flag could overtake the other seven waves' data (round 4: the send / recv ring of the world-8
// test and the TP block's all-reduces failed intermittently once eight processes time-sliced one GPU).
// (A wait, not a fence: once a wave's stores have left it — vmcnt(0): written through for the uncached destination, or sitting in
// this XCD's L2 — the ONE system-scope release fence lane 0 executes behind the barrier (publish: buffer_wbl2 + wait) covers them,
// because the L2 it writes back is the one every wave of this workgroup stores through: a fence per wave would repeat that
// write-back eight times per workgroup and phase.)
__device__ __forceinline__ void release_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Element range [e0, e1) of workgroup b in a `len`-element range. The partition is in 16-BYTE units whatever vector width a
// rank's kernel uses (the width follows the alignment of that rank's own pointers; the flags are per workgroup, so writer and
// reader of a call must cut it identically): boundaries are multiples of 16 / sizeof(T) elements, the last range ends at len.
template <typename T> __device__ __forceinline__ void elem_range(long len, int b, long &e0, long &e1) {
    constexpr long PU = 16 / (long)sizeof(T);
    const long units = (len + PU - 1) / PU;
    e0 = units * b / kDGrid * PU;
    e1 = units * (b + 1) / kDGrid * PU;
    e0 = e0 < len ? e0 : len;
    e1 = e1 < len ? e1 : len;
}

template <typename T, int VEC> struct alignas(sizeof(T) * VEC) Pack { T v[VEC]; };

// copy elements [e0, e1) (e0 a multiple of VEC; a last partial vector goes element by element)
template <typename T, int VEC>
__device__ __forceinline__ void copy_range(const T *src, T *dst, long e0, long e1) {
    for (long e = e0 + (long)threadIdx.x * VEC; e < e1; e += (long)kDThreads * VEC) {
        if (VEC > 1 && e + VEC <= e1) {
            *reinterpret_cast<Pack<T, VEC> *>(dst + e) = *reinterpret_cast<const Pack<T, VEC> *>(src + e);
        } else {
            for (long i = e; i < e1 && i < e + VEC; ++i)
                dst[i] = src[i];
        }
    }
}

// Every thread; ends with a __syncthreads(). Thread t < (world - 1) * G waits for ONE word: flags fa (or, when given, fb) of peer
// t / G, workgroup t % G in MY ctrl to reach `want`. Used where a workgroup is about to overwrite box space that workgroups OTHER
// than its own partner may still be reading: the ranges of two calls differ when their lengths do, so the per-workgroup flag of
// the partner proves nothing about the neighbours' ranges (round 4: the last, shorter piece of a multi-piece broadcast overwrote
// what slower workgroups of a peer were still copying out; eight processes time-slicing one GPU made it visible).
__device__ __forceinline__ void wait_all_workgroups(const DirectArgs &a, unsigned (*fa)[kDGrid], unsigned (*fb)[kDGrid], unsigned want,
                                                    int only_peer) {
    const int n = a.world, me = a.rank, t = threadIdx.x;
    const int npeers = only_peer >= 0 ? 1 : n - 1;
    if (want != 0 && t < npeers * kDGrid) {
        const int p = only_peer >= 0 ? only_peer : (me + 1 + t / kDGrid) % n, wb = t % kDGrid;
        if (__hip_atomic_load(&ctrl_of(a, me)->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
            const long long t0 = wall_clock64();
            for (;;) {
                if ((int)(__hip_atomic_load(&fa[p][wb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - want) >= 0)
                    break;
                if (fb && (int)(__hip_atomic_load(&fb[p][wb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - want) >= 0)
                    break;
                __builtin_amdgcn_s_sleep(4);
                if (wall_clock64() - t0 > a.timeout_ticks) {
                    __hip_atomic_store(&ctrl_of(a, me)->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
        }
    }
    __syncthreads();
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
    // boxes of parity par were last used by call s - 2: EVERY workgroup of every peer has left it once it published call s - 1
    wait_all_workgroups(a, ctrl_of(a, me)->flagS, ctrl_of(a, me)->flagG, s - 1, -1);
    auto slice_len = [&](int j) {
        if (fixed_len >= 0)
            return fixed_len;
        const long rest = count - (long)j * slice;
        return rest < 0 ? 0 : (rest < slice ? rest : slice);
    };
    // ---- phase 1: push slice j of my x into inbox[par][me] of rank j (all peers inside the loop: every link busy)
    {
        long maxe = 0;
        for (int j = 0; j < n; ++j)
            if (j != me) {
                long e0, e1;
                elem_range<T>(slice_len(j), b, e0, e1);
                maxe = e1 - e0 > maxe ? e1 - e0 : maxe;
            }
        for (long i = (long)threadIdx.x * VEC; i < maxe; i += (long)kDThreads * VEC)
            for (int d = 1; d < n; ++d) {
                const int j = (me + d) % n;
                long e0, e1;
                elem_range<T>(slice_len(j), b, e0, e1);
                const long e = e0 + i;
                if (e >= e1)
                    continue;
                const T *src = x + (long)j * slice;
                T *dst = (T *)inbox_of(a, j, par, me);
                if (VEC > 1 && e + VEC <= e1) {
                    *reinterpret_cast<Pack<T, VEC> *>(dst + e) = *reinterpret_cast<const Pack<T, VEC> *>(src + e);
                } else {
                    for (long q = e; q < e1 && q < e + VEC; ++q)
                        dst[q] = src[q];
                }
            }
    }
    release_stores();
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
        long e0, e1;
        elem_range<T>(slice_len(me), b, e0, e1);
        const T *mine = x + (long)me * slice;
        T *out = mode == 0 ? y + (long)me * slice : y;
        for (long e = e0 + (long)threadIdx.x * VEC; e < e1; e += (long)kDThreads * VEC) {
            const int lim = (int)(e1 - e < VEC ? e1 - e : VEC);
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
        release_stores();
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
            long e0, e1;
            elem_range<T>(slice_len(src), b, e0, e1);
            copy_range<T, VEC>((const T *)gbox_of(a, me, par, src), y + (long)src * slice, e0, e1);
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
    wait_all_workgroups(a, ctrl_of(a, me)->flagS, ctrl_of(a, me)->flagG, s - 1, -1); // (see direct_reduce_kernel)
    long e0, e1;
    elem_range<T>(len, b, e0, e1);
    for (long e = e0 + (long)threadIdx.x * VEC; e < e1; e += (long)kDThreads * VEC) {
        if (VEC > 1 && e + VEC <= e1) {
            const Pack<T, VEC> v = *reinterpret_cast<const Pack<T, VEC> *>(x + e);
            for (int d = 1; d < n; ++d)
                *reinterpret_cast<Pack<T, VEC> *>(gbox_of(a, (me + d) % n, par, me) + e) = v;
            *reinterpret_cast<Pack<T, VEC> *>(y + (long)me * ystride + e) = v;
        } else {
            for (long q = e; q < e1 && q < e + VEC; ++q) {
                const char v = x[q];
                for (int d = 1; d < n; ++d)
                    gbox_of(a, (me + d) % n, par, me)[q] = v;
                y[(long)me * ystride + q] = v;
            }
        }
    }
    release_stores();
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
        copy_range<T, VEC>(gbox_of(a, me, par, src), y + (long)src * ystride, e0, e1);
    }
    __syncthreads();
    if (threadIdx.x == 0)
        a.local->seq[b] = s;
}

template <int VEC>
__global__ __launch_bounds__(kDThreads) void direct_broadcast_kernel(DirectArgs a, const char *x, char *y, long len, int root) {
    using T = char;
    const int b = blockIdx.x, me = a.rank, n = a.world;
    __shared__ unsigned s_sh, last_sh;
    if (threadIdx.x == 0) {
        s_sh = a.local->bseq[b] + 1;
        last_sh = a.local->broot_last[b];
    }
    __syncthreads();
    const unsigned s = s_sh;
    long e0, e1;
    elem_range<T>(len, b, e0, e1);
    if (me == root) {
        // credit: my previous broadcast has been copied out of every peer's bbox[me] by ALL of the peer's workgroups (its length,
        // and with it the ranges, may have been different)
        wait_all_workgroups(a, ctrl_of(a, me)->ackB, nullptr, last_sh, -1);
        for (long e = e0 + (long)threadIdx.x * VEC; e < e1; e += (long)kDThreads * VEC) {
            if (VEC > 1 && e + VEC <= e1) {
                const Pack<T, VEC> v = *reinterpret_cast<const Pack<T, VEC> *>(x + e);
                for (int d = 1; d < n; ++d)
                    *reinterpret_cast<Pack<T, VEC> *>(bbox_of(a, (me + d) % n, root) + e) = v;
                if (y != x)
                    *reinterpret_cast<Pack<T, VEC> *>(y + e) = v;
            } else {
                for (long q = e; q < e1 && q < e + VEC; ++q) {
                    const char v = x[q];
                    for (int d = 1; d < n; ++d)
                        bbox_of(a, (me + d) % n, root)[q] = v;
                    y[q] = v;
                }
            }
        }
        release_stores();
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
        copy_range<T, VEC>(bbox_of(a, me, root), y, e0, e1);
        __syncthreads(); // (a wave that arrives here has issued its stores to y, i.e. its loads from the box have returned)
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
    if (threadIdx.x == 0)
        s_sh = a.local->sseq[peer][b] + 1;
    __syncthreads();
    const unsigned s = s_sh;
    // credit: the previous message has left the peer's pbox[me] — ALL of the peer's workgroups have copied their ranges out
    wait_all_workgroups(a, ctrl_of(a, me)->ackP, nullptr, s - 1, peer);
    long e0, e1;
    elem_range<T>(len, b, e0, e1);
    copy_range<T, VEC>(x, pbox_of(a, peer, me), e0, e1);
    release_stores();
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
    long e0, e1;
    elem_range<T>(len, b, e0, e1);
    copy_range<T, VEC>(pbox_of(a, me, peer), y, e0, e1);
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
    // job: a hash of what the launcher gives every rank of ONE job (INFINI_ROCM_JOB_ID, else torchrun's TORCHELASTIC_RUN_ID, else
    // MASTER_ADDR:MASTER_PORT), 0 when the environment holds none of them — a file with another job's value is a stale file.
    // proc: a random token drawn once per process — "this file was written by ME" without comparing PIDs, which mean nothing
    // across PID namespaces (containers sharing the rendezvous directory).
    unsigned long long job, proc;
    hipIpcMemHandle_t handle;
};

static unsigned long long fnv1a(const std::string &v) {
    unsigned long long h = 1469598103934665603ull;
    for (unsigned char c : v)
        h = (h ^ c) * 1099511628211ull;
    return h ? h : 1ull;
}
static unsigned long long job_nonce() {
    if (const char *v = std::getenv("INFINI_ROCM_JOB_ID"))
        return fnv1a(std::string("id:") + v);
    if (const char *v = std::getenv("TORCHELASTIC_RUN_ID"))
        return fnv1a(std::string("run:") + v);
    const char *a = std::getenv("MASTER_ADDR"), *p = std::getenv("MASTER_PORT");
    if (a && p)
        return fnv1a(std::string("master:") + a + ":" + p);
    return 0ull;
}
static unsigned long long proc_token() {
    static const unsigned long long tok = [] {
        unsigned long long v = 0;
        std::ifstream ur("/dev/urandom", std::ios::binary);
        ur.read((char *)&v, sizeof(v));
        v ^= (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() * 0x9e3779b97f4a7c15ull ^ (unsigned long long)getpid();
        return v ? v : 1ull;
    }();
    return tok;
}

static bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

int direct_init(infiniRocmRuntime *rt, const char *name, int world, int rank) {
    IROCM_CHECK_ARG(world >= 1 && world <= kDMaxRanks, "direct transport: world size %d (1..%d: one xGMI node)", world, kDMaxRanks);
    IROCM_CHECK_ARG(rt->dcomm == nullptr, "direct communicator already initialised");
    IROCM_HIP(hipSetDevice(rt->device));
    auto *dc = new DirectComm();
    const char *cap_env = std::getenv("INFINI_ROCM_DIRECT_CAP_MB");
    const char *to_env = std::getenv("INFINI_ROCM_DIRECT_TIMEOUT_S");
    size_t cap = (size_t)(cap_env ? std::max(1, std::atoi(cap_env)) : 8) << 20;
    // Default 600 s: a peer that is merely slow (weight loading, tune(), first-iteration skew) must be waited for like RCCL would;
    // the limit exists so that a DEAD peer ends as an error (reported by the next runtime_sync) instead of a hung GPU. Tests and
    // bench.py set a short one.
    const double timeout_s = to_env ? std::atof(to_env) : 600.0;
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
        memcpy(me.magic, "IROCMXG2", 8);
        me.world = world, me.rank = rank, me.grid = kDGrid, me.pid = (int)getpid(), me.cap = cap;
        me.job = job_nonce(), me.proc = proc_token();
        if ((e = hipIpcGetMemHandle(&me.handle, dc->block)) != hipSuccess)
            return fail(INFINI_ROCM_HIP_ERROR, std::string("direct transport: hipIpcGetMemHandle failed: ") + hipGetErrorString(e) +
                                                   " (is HSA_ENABLE_IPC_MODE_LEGACY=0 exported?)");
        auto path = [&](int r) { return std::string("./") + name + "_xgmi_" + std::to_string(r) + ".bin"; };
        dc->my_file = path(rank);
        (void)unlink(dc->my_file.c_str()); // a file of the same name left by a crashed job must never be read as mine
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
            // "not (yet) the file of my peer" — retried until the time limit, never an immediate failure: a short read (the peer
            // unlinked its own stale file between my stat() and my read, or the read raced a replace), a foreign magic, another
            // job's nonce, and — only when the launcher gave no nonce — a writer PID that no longer exists in MY PID namespace.
            const bool whole = ifs.gcount() == (std::streamsize)sizeof(peer) && memcmp(peer.magic, "IROCMXG2", 8) == 0;
            const bool other_job = whole && peer.job != me.job;
            const bool dead_writer = whole && me.job == 0ull && peer.proc != me.proc && kill((pid_t)peer.pid, 0) != 0 && errno == ESRCH;
            if (!whole || other_job || dead_writer) {
                if (std::chrono::steady_clock::now() > begin + std::chrono::seconds(120))
                    return fail(INFINI_ROCM_RCCL_ERROR, "direct transport: " + path(r) +
                                                            (!whole ? " never became a complete rendezvous record"
                                                                    : other_job ? " belongs to another job (stale file?)"
                                                                                : " was written by a process that no longer exists"));
                std::this_thread::sleep_for(std::chrono::milliseconds(50));
                --r;
                continue;
            }
            if (peer.proc == me.proc)
                return fail(INFINI_ROCM_UNSUPPORTED, "direct transport: rank " + std::to_string(r) + " lives in THIS process; an IPC handle cannot be "
                                                     "opened by its exporter (one process per rank; ranks may share a device)");
            if (peer.world != world || peer.rank != r || peer.grid != kDGrid || peer.cap != cap)
                return fail(INFINI_ROCM_RCCL_ERROR, "direct transport: " + path(r) + " does not match this job (different settings)");
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
    rt->dcomm_dirty = 0;
    IROCM_HIP(hipMemcpy(&err, &((DirectCtrl *)dc->block)->error, sizeof(err), hipMemcpyDeviceToHost));
    if (err) {
        // reported ONCE, then cleared: the kernels skip every wait while the word is set (one time limit per rank, not one per flag),
        // so leaving it set would turn every later collective of this communicator into garbage without a wait (round-4 advisor)
        const unsigned zero = 0;
        IROCM_HIP(hipMemcpy(&((DirectCtrl *)dc->block)->error, &zero, sizeof(zero), hipMemcpyHostToDevice));
        IROCM_FAIL(INFINI_ROCM_RCCL_ERROR, "direct transport: a peer did not arrive within the time limit (results of the collectives "
                                           "since the last sync are undefined; the error is now cleared)");
    }
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
    rt->dcomm_dirty = 1; // (infini_rocm_runtime_sync reads the error word once after the stream has drained)
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
    rt->dcomm_dirty = 1; // (infini_rocm_runtime_sync reads the error word once after the stream has drained)
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
    rt->dcomm_dirty = 1; // (infini_rocm_runtime_sync reads the error word once after the stream has drained)
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
    rt->dcomm_dirty = 1; // (infini_rocm_runtime_sync reads the error word once after the stream has drained)
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
    rt->dcomm_dirty = 1; // (infini_rocm_runtime_sync reads the error word once after the stream has drained)
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
    rt->dcomm_dirty = 1; // (infini_rocm_runtime_sync reads the error word once after the stream has drained)
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
