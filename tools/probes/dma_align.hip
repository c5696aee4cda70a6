// Probe: does a 16-byte LDS-DMA piece (global_load_lds_dwordx4) accept global addresses that are only 8- / 4- / 2-byte aligned?
// (A pointwise convolution over 14 x 14 planes has 392-byte rows: its 16-byte runs start on 8-byte boundaries.)
// hipcc --offload-arch=gfx950 -O2 tools/probes/dma_align.hip -o gpurun_out/dma_align && gpurun_out/dma_align
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void probe(const unsigned short *src, unsigned short *dst, int shift_elems) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 8];
    const int lane = threadIdx.x;
    // lane i fetches 8 elements starting at element shift + 8 * i  ->  LDS bytes [16 i, 16 i + 16)
    const unsigned short *g = src + shift_elems + 8 * lane;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int e = 0; e < 8; ++e)
        dst[lane * 8 + e] = lds[lane * 8 + e];
}

int main() {
    const int n = 4096;
    std::vector<unsigned short> h(n);
    for (int i = 0; i < n; ++i) h[i] = (unsigned short)i;
    unsigned short *src, *dst;
    hipMalloc(&src, n * 2);
    hipMalloc(&dst, 512 * 2);
    hipMemcpy(src, h.data(), n * 2, hipMemcpyHostToDevice);
    for (int shift : {0, 4, 2, 1, 12, 3}) { // 0: 16-byte aligned, 4: 8-byte, 2: 4-byte, 1: 2-byte
        hipMemset(dst, 0xff, 512 * 2);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, dst, shift);
        hipError_t e = hipDeviceSynchronize();
        std::vector<unsigned short> out(512);
        hipMemcpy(out.data(), dst, 512 * 2, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 512; ++i) bad += out[i] != (unsigned short)(i + shift);
        printf("shift %2d elements (%2d-byte aligned): %s, %d / 512 elements wrong (first: got %u want %u)\n", shift,
               (shift * 2) % 16 == 0 ? 16 : ((shift * 2) % 8 == 0 ? 8 : ((shift * 2) % 4 == 0 ? 4 : 2)), hipGetErrorString(e), bad,
               out[0], (unsigned)shift);
    }
    return 0;
}
