// Feasibility probe for the hand-written xGMI / IPC transport (csrc/comm_direct.hip): two PROCESSES on ONE GPU.
//  1. which allocation flavours can be exported with hipIpcGetMemHandle (uncached / fine-grained / plain hipMalloc);
//  2. a kernel of process A spinning (bounded) on a flag that a kernel of process B sets after writing a payload into A's
//     memory through the IPC mapping: do the two kernels run concurrently, is the payload visible after a system-scope
//     acquire, how long does the hand-off take.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/ipc_probe.hip -o /tmp/ipc_probe && HSA_ENABLE_IPC_MODE_LEGACY=0 /tmp/ipc_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/wait.h>
#include <unistd.h>

#define CK(e)                                                                                                    \
    do {                                                                                                         \
        hipError_t _e = (e);                                                                                     \
        if (_e != hipSuccess) {                                                                                  \
            fprintf(stderr, "[pid %d] %s -> %s (line %d)\n", getpid(), #e, hipGetErrorString(_e), __LINE__);     \
            exit(2);                                                                                             \
        }                                                                                                        \
    } while (0)

struct Box {
    unsigned flag;
    unsigned ack;
    unsigned pad[62];
    unsigned data[1 << 20];
};

__global__ void producer(Box *peer, unsigned seq) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (1 << 20); i += gridDim.x * blockDim.x)
        peer->data[i] = seq * 1000003u + i;
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&peer->flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ void consumer(Box *mine, unsigned seq, unsigned want_arrivals, unsigned *result, long long timeout_ticks) {
    __shared__ int ok;
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        ok = 1;
        while (__hip_atomic_load(&mine->flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want_arrivals) {
            __builtin_amdgcn_s_sleep(16);
            if (wall_clock64() - t0 > timeout_ticks) {
                ok = 0;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        result[2] = (unsigned)((wall_clock64() - t0) / 100); // microseconds at 100 MHz
    }
    __syncthreads();
    unsigned bad = 0;
    if (ok)
        for (int i = threadIdx.x; i < (1 << 20); i += blockDim.x)
            bad += mine->data[i] != seq * 1000003u + i;
    atomicAdd(&result[0], bad);
    if (threadIdx.x == 0)
        result[1] = ok;
}

int main() {
    int to_child[2], to_parent[2];
    pipe(to_child);
    pipe(to_parent);
    const pid_t pid = fork(); // before any HIP call
    if (pid == 0) { // ---- child: the producer (rank 1)
        hipIpcMemHandle_t h;
        int flavour;
        if (read(to_child[0], &flavour, sizeof(flavour)) != sizeof(flavour) || read(to_child[0], &h, sizeof(h)) != sizeof(h))
            exit(3);
        CK(hipSetDevice(0));
        void *p = nullptr;
        CK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        char c = 'o';
        write(to_parent[1], &c, 1); // opened
        for (unsigned seq = 1; seq <= 3; ++seq) {
            read(to_child[0], &c, 1); // parent launched its consumer
            usleep(seq == 1 ? 200000 : 1000);
            hipLaunchKernelGGL(producer, dim3(64), dim3(256), 0, 0, (Box *)p, seq);
            CK(hipDeviceSynchronize());
        }
        CK(hipIpcCloseMemHandle(p));
        exit(0);
    }
    CK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    Box *box = nullptr;
    hipIpcMemHandle_t h;
    int flavour = -1;
    const unsigned flags[3] = {hipDeviceMallocUncached, hipDeviceMallocFinegrained, 0};
    const char *names[3] = {"uncached", "finegrained", "hipMalloc"};
    for (int f = 0; f < 3 && flavour < 0; ++f) {
        hipError_t e = flags[f] ? hipExtMallocWithFlags((void **)&box, sizeof(Box), flags[f]) : hipMalloc((void **)&box, sizeof(Box));
        if (e != hipSuccess) {
            printf("alloc %s: %s\n", names[f], hipGetErrorString(e));
            (void)hipGetLastError();
            continue;
        }
        e = hipIpcGetMemHandle(&h, box);
        printf("alloc %s ok; hipIpcGetMemHandle: %s\n", names[f], hipGetErrorString(e));
        if (e == hipSuccess)
            flavour = f;
        else {
            (void)hipGetLastError();
            (void)hipFree(box);
        }
    }
    if (flavour < 0)
        return 1;
    CK(hipMemset(box, 0, sizeof(Box)));
    CK(hipDeviceSynchronize());
    write(to_child[1], &flavour, sizeof(flavour));
    write(to_child[1], &h, sizeof(h));
    char c;
    read(to_parent[0], &c, 1);
    printf("child opened the handle (%s memory)\n", names[flavour]);
    unsigned *res = nullptr;
    CK(hipMalloc((void **)&res, 16));
    for (unsigned seq = 1; seq <= 3; ++seq) {
        CK(hipMemset(res, 0, 16));
        hipLaunchKernelGGL(consumer, dim3(1), dim3(256), 0, 0, box, seq, seq * 64u, res, 300000000ll); // 3 s
        c = 'g';
        write(to_child[1], &c, 1);
        CK(hipDeviceSynchronize());
        unsigned host[4];
        CK(hipMemcpy(host, res, 16, hipMemcpyDeviceToHost));
        printf("round %u: flag seen %u, wrong words %u, consumer waited %u us\n", seq, host[1], host[0], host[2]);
    }
    int st = 0;
    waitpid(pid, &st, 0);
    printf("child exit %d\n", WEXITSTATUS(st));
    return 0;
}
