#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const unsigned short *x, unsigned nbytes, const int *offs, unsigned short *out, int n) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)nbytes, 0x00020000);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, offs[i], 0, 0);
    for (int j = 0; j < 4; ++j) { out[i * 8 + 2 * j] = v[j] & 0xffff; out[i * 8 + 2 * j + 1] = v[j] >> 16; }
}
__global__ void bw(const unsigned short *x, unsigned nbytes, int shift, unsigned *sink, long nchunks) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)nbytes, 0x00020000);
    unsigned acc = 0;
    for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += (long)gridDim.x * blockDim.x) {
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(c * 16 + shift), 0, 0);
        acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x12345) sink[0] = acc;
}
int main() {
    const int N = 1 << 20;
    std::vector<unsigned short> h(N);
    for (int i = 0; i < N; ++i) h[i] = (unsigned short)(i * 7 + 1);
    unsigned short *dx, *dout; int *doffs;
    hipMalloc(&dx, N * 2 + 4096); hipMemcpy(dx, h.data(), N * 2, hipMemcpyHostToDevice);
    std::vector<int> offs = {0, 2, 4, 6, 10, 14, 18, 30, 126, 2 * N - 16, 2 * N - 14, 2 * N - 2, 2 * N, -2, -14, -16, -32, 2 * N - 6};
    int n = offs.size();
    hipMalloc(&doffs, n * 4); hipMemcpy(doffs, offs.data(), n * 4, hipMemcpyHostToDevice);
    hipMalloc(&dout, n * 16);
    for (unsigned nb : {(unsigned)(2 * N), (unsigned)(2 * N + 2)}) {
        probe<<<1, 64>>>(dx, nb, doffs, dout, n);
        std::vector<unsigned short> o(n * 8);
        hipMemcpy(o.data(), dout, n * 16, hipMemcpyDeviceToHost);
        printf("num_records=%u\n", nb);
        for (int i = 0; i < n; ++i) {
            printf(" off %8d:", offs[i]);
            int ok = 1;
            for (int j = 0; j < 8; ++j) {
                long e = offs[i] / 2 + j; unsigned short want = (e >= 0 && e < N) ? h[e] : 0;
                printf(" %5u%s", o[i * 8 + j], o[i * 8 + j] == want ? "" : "!");
                ok &= o[i * 8 + j] == want;
            }
            printf("  %s\n", ok ? "ok" : "DIFF");
        }
    }
    // bandwidth: aligned vs +2 vs +8 shift over 512 MB
    const long BYTES = 512l << 20;
    unsigned short *big; unsigned *sink; hipMalloc(&big, BYTES + 4096); hipMalloc(&sink, 4); hipMemset(big, 1, BYTES + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shift : {0, 2, 4, 8, 6}) {
        bw<<<256 * 8, 256>>>(big, (unsigned)BYTES, shift, sink, BYTES / 16 - 1);
        hipEventRecord(e0);
        for (int it = 0; it < 5; ++it) bw<<<256 * 8, 256>>>(big, (unsigned)BYTES, shift, sink, BYTES / 16 - 1);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("shift %d: %.1f GB/s\n", shift, 5.0 * BYTES / ms / 1e6);
    }
    return 0;
}
