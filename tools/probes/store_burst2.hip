// Probe 2: the persistent GEMM's epilogue store pattern in isolation (see store_burst.hip): wave (wr, wc) of 8 owns the 128 x 64
// sub-tile at (wr * 128, wc * 64) of a 256 x 256 x 16-bit tile; store step i writes rows i*16 + l15, two 64-byte column halves.
// VALU = dependent vector ops on the store data between stores; DELAY = s_sleep units before wave row 1 starts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int VALU, int SADDR, int MAP> __global__ __launch_bounds__(512) void burst(char *c, int ldc2, long long *cycles, int tiles_x, int delay) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = w >> 2, wc = w & 3, l15 = lane & 15, g4 = lane >> 4;
    const int tile = blockIdx.x;
    char *base = c + ((long)(tile / tiles_x) * 256 * ldc2) + (long)(tile % tiles_x) * 512;
    u32x4 v = {(unsigned)lane, (unsigned)w, 3u, 4u};
    const bool odd = g4 & 1;
    unsigned voff[2];
    // MAP 0: the epilogue's lane -> address map (lane = g4 * 16 + l15: 16 consecutive lanes walk DOWN 16 rows)
    // MAP 1: row = lane / 4, 16-byte chunk = lane % 4 (4 consecutive lanes = 64 contiguous bytes), two stores = the two 64-byte halves
    // MAP 2: row = lane / 8 (+ 8 for the second store), chunk = lane % 8 (8 consecutive lanes = one 128-byte line)
    for (int jp = 0; jp < 2; ++jp) {
        if (MAP == 0) voff[jp] = l15 * ldc2 + (wc * 64 + (jp * 2 + (odd ? 1 : 0)) * 16 + (g4 & ~1) * 4) * 2;
        else if (MAP == 1) voff[jp] = (lane >> 2) * ldc2 + wc * 128 + jp * 64 + (lane & 3) * 16;
        else voff[jp] = ((lane >> 3) + jp * 8) * ldc2 + wc * 128 + (lane & 7) * 16;
    }
    char *sbase = base + (long)(wr * 128) * ldc2;
    if (wr == 1)
        for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(8);
    const long long t0 = clock64();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
#pragma unroll
            for (int q = 0; q < VALU; ++q)
                v[q & 3] = v[q & 3] * 3u + v[(q + 1) & 3];
            if (SADDR) {
                asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(voff[jp]), "v"(v), "s"(sbase) : "memory");
            } else {
                char *p = sbase + voff[jp];
                asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
            }
        }
        sbase += 16l * ldc2;
    }
    const long long t1 = clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t2 = clock64();
    if (lane == 0) {
        cycles[(blockIdx.x * 8 + w) * 2] = t1 - t0;
        cycles[(blockIdx.x * 8 + w) * 2 + 1] = t2 - t0;
    }
}

template <int VALU, int SADDR, int MAP> static void run(char *c, long long *cyc, int grid, int n, int delay) {
    const int tiles_x = n / 256;
    for (int it = 0; it < 2; ++it) {
        hipLaunchKernelGGL((burst<VALU, SADDR, MAP>), dim3(grid), dim3(512), 0, 0, c, n * 2, cyc, tiles_x, delay);
        hipDeviceSynchronize();
    }
    std::vector<long long> h(grid * 16);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double is0 = 0, is1 = 0, dn0 = 0, dn1 = 0;
    for (int b = 0; b < grid; ++b)
        for (int w = 0; w < 8; ++w) {
            (w < 4 ? is0 : is1) += (double)h[(b * 8 + w) * 2] / (grid * 4);
            (w < 4 ? dn0 : dn1) += (double)h[(b * 8 + w) * 2 + 1] / (grid * 4);
        }
    printf("grid %3d n %5d  map %d valu %2d saddr %d delay %2d: row 0 issue %6.0f done %6.0f | row 1 issue %6.0f done %6.0f cycles\n", grid, n, MAP, VALU, SADDR,
           delay, is0, dn0, is1, dn1);
}

int main() {
    char *c; long long *cyc;
    hipMalloc(&c, 8192l * 8192 * 2 + 4096);
    hipMalloc(&cyc, 256 * 16 * 8);
    for (int grid : {256, 64}) {
        for (int n : {4096, 2048, 3072}) {
            if (grid * 256 * 256 > (long)n * n * (n == 3072 ? 6 : 1) ) continue;
            run<10, 1, 0>(c, cyc, grid, n, 0);
            run<10, 1, 1>(c, cyc, grid, n, 0);
            run<10, 1, 2>(c, cyc, grid, n, 0);
            run<10, 1, 0>(c, cyc, grid, n, 8);
            run<10, 1, 1>(c, cyc, grid, n, 8);
            run<10, 1, 2>(c, cyc, grid, n, 8);
        }
    }
    return 0;
}
