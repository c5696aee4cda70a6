// Probe: how fast can ONE CU retire a 256 x 256 x 16-bit output tile (128 KiB) with global_store_dwordx4, as a function of the
// address pattern of one wave-wide store (how many rows x how many bytes per row), with the rest of the chip quiet or equally busy?
// The persistent GEMM's epilogue takes ~10 k cycles per tile (13 B/clk/CU); is that the CU's store path or the pattern?
// hipcc --offload-arch=gfx950 -O2 tools/probes/store_burst.hip -o gpurun_out/store_burst && gpurun_out/store_burst
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 8 waves; wave w owns rows [32 w, 32 w + 32) of the workgroup's 256 x 256 tile of a row-major [4096][4096] 16-bit matrix.
// ROWS = rows one store instruction touches (64 lanes x 16 B = 1 KiB spread over ROWS rows of 1024 / ROWS bytes each).
template <int ROWS, int WAVES> __global__ __launch_bounds__(WAVES * 64) void burst(char *c, int reps, long long *cycles, int tiles_x) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int tile = blockIdx.x;
    char *base = c + ((long)(tile / tiles_x) * 256 * 8192) + (long)(tile % tiles_x) * 512;
    constexpr int RB = 1024 / ROWS;          // bytes per row per instruction
    constexpr int LPR = RB / 16;             // lanes per row
    constexpr int STRIPS = 512 / RB;         // instructions to cover a 512-byte tile row span
    constexpr int ROWS_PER_WAVE = 256 / WAVES;
    constexpr int NS = ROWS_PER_WAVE * 512 / 1024;
    u32x4 v = {(unsigned)lane, (unsigned)w, 3u, 4u};
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int rowblk = j / STRIPS, strip = j % STRIPS;
            const int row = w * ROWS_PER_WAVE + rowblk * ROWS + lane / LPR;
            char *p = base + (long)row * 8192 + strip * RB + (lane % LPR) * 16;
            asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
        }
        v.x += 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    if (lane == 0)
        cycles[blockIdx.x * WAVES + w] = t1 - t0;
}

template <int ROWS, int WAVES> static void run(char *c, long long *cyc, int grid, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int tiles_x = 16;
    hipLaunchKernelGGL((burst<ROWS, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, c, reps, cyc, tiles_x);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((burst<ROWS, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, c, reps, cyc, tiles_x);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid * WAVES);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    long long mx = 0; double avg = 0;
    for (auto x : h) { mx = x > mx ? x : mx; avg += (double)x / h.size(); }
    const double bytes = 131072.0 * reps;
    printf("grid %3d waves %d  %2d rows x %4d B per store: %8.0f cycles per tile (avg over waves; max %8.0f)  %5.1f B/clk/CU   kernel %7.1f us  %6.2f TB/s\n",
           grid, WAVES, ROWS, 1024 / ROWS, avg / reps, (double)mx / reps, bytes / avg, ms * 1e3, bytes * grid / (ms * 1e-3) / 1e12);
}

int main() {
    char *c; long long *cyc;
    hipMalloc(&c, 4096l * 8192 + 4096);
    hipMalloc(&cyc, 256 * 8 * 8);
    for (int grid : {256, 64, 8}) {
        for (int reps : {1, 8}) {
            printf("-- reps %d\n", reps);
            run<16, 8>(c, cyc, grid, reps);
            run<8, 8>(c, cyc, grid, reps);
            run<4, 8>(c, cyc, grid, reps);
            run<2, 8>(c, cyc, grid, reps);
            run<16, 4>(c, cyc, grid, reps);
            run<2, 4>(c, cyc, grid, reps);
        }
    }
    return 0;
}
