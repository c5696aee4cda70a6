// Shared pieces of the 256x256x64 GEMM kernels (gemm256.hip, gemm256p_kernel.h: 16x16x32 MFMA).
#pragma once
#include "gemm_common.h"
#include <type_traits>
#include <utility>

namespace irocm {
namespace g256 {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int OPER_BYTES = 256 * 64 * 2;  // 32 KiB per operand tile
constexpr int BUF_BYTES = 2 * OPER_BYTES; // A | B
constexpr int LDS_BYTES = 2 * BUF_BYTES;  // 128 KiB

template <typename F, int... I> __device__ __forceinline__ void sfor_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void sfor(F &&f) {
    sfor_impl(f, std::make_integer_sequence<int, N>{});
}

template <int OFF> __device__ __forceinline__ s16x8_t lds_read_b128(unsigned addr) {
    s16x8_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
template <int OFF> __device__ __forceinline__ s16x4_t lds_read_tr_b64(unsigned addr) {
    s16x4_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
// One MFMA operand fragment (8 x 16-bit). K-major: one ds_read_b128. M/N-major: two transpose reads
// whose halves are only joined into one 128-bit value AFTER the lgkmcnt wait (a v_mov issued between the
// asm read and the wait would copy stale registers: hipcc does not know the asm is a load).
template <bool KMAJOR> struct Frag;
template <> struct Frag<true> {
    s16x8_t v;
    __device__ __forceinline__ s16x8_t get() const { return v; }
};
template <> struct Frag<false> {
    s16x4_t lo, hi;
    __device__ __forceinline__ s16x8_t get() const {
        return s16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
};

// ds_bpermute_b32 (dst lane L <- src lane addr[L] / 4) as opaque asm: hipcc then inserts no lgkmcnt(0) of its own before the first
// use and the caller waits with a COUNT (wait_lgkm<N>: results of LDS instructions return in order), i.e. can keep the next
// exchange in flight under the stores of this one. The result must not be read before that wait.
// The counted form rests on two things the compiler does not know about: it must not touch the result registers (copy, coalesce,
// spill) between the exchange and the wait, and it must issue no LDS / SMEM operation of its own in between. Verified (ISA
// read + the bit-exact epilogue tests of tests/test_gpu_matmul.py / test_gpu_nn.py) for the ROCm 7.2 compiler only: any other
// compiler — or -DIROCM_SAFE_BPERMUTE=1 — gets the builtin, whose waits the compiler manages itself (slower, always right).
#if !defined(IROCM_SAFE_BPERMUTE)
#if defined(__clang_major__) && __clang_major__ == 22 && defined(HIP_VERSION_MAJOR) && HIP_VERSION_MAJOR == 7 && HIP_VERSION_MINOR == 2
#define IROCM_SAFE_BPERMUTE 0
#else
#define IROCM_SAFE_BPERMUTE 1
#endif
#endif
__device__ __forceinline__ unsigned lds_bpermute(int addr, unsigned v) {
#if IROCM_SAFE_BPERMUTE
    return (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)v);
#else
    unsigned r;
    asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(r) : "v"(addr), "v"(v));
    return r;
#endif
}
template <int N> __device__ __forceinline__ void wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void wait_lgkm0() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void fence_sched() { __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ void barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// ---- staging (LDS-DMA) -----------------------------------------------------------------------
// K-major operand: 32 pieces of 8 rows; wave w issues pieces w*4 .. w*4+3.
__device__ __forceinline__ void offs_k(unsigned (&off)[4], long ld, int row0, int rows, int w, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = w * 4 + i;
        const int r = piece * 8 + (lane >> 3);
        const int c_log = (lane & 7) ^ ((r >> 1) & 7);
        int gr = row0 + r;
        gr = gr < rows ? gr : rows - 1; // rows beyond the matrix re-read its last row; never stored
        off[i] = (unsigned)(((long)gr * ld + c_log * 8) * 2);
    }
}
// M/N-major operand: image [64 k][256 cols] (512-B rows), 32 pieces of 2 k-rows.
__device__ __forceinline__ int mn_f(int kr) { return (kr & 3) | (((kr >> 3) & 1) << 2); }
__device__ __forceinline__ void offs_mn(unsigned (&off)[4], long ld, int col0, int cols, int w, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = w * 4 + i;
        const int kr = piece * 2 + (lane >> 5);
        const int c_log = (lane & 31) ^ (mn_f(kr) << 1);
        int gc = col0 + c_log * 8;
        gc = gc <= cols - 8 ? gc : cols - 8;
        off[i] = (unsigned)(((long)kr * ld + gc) * 2);
    }
}
// conv mode: the B tile's columns are pixel slots (image, pixel) of an NCHW activation; k-row kr of the tile is channel k0 + kr.
// Byte offset of the lane's 16-byte run from X: ((img * C + kr) * HW + pix) * 2. A run that starts inside a plane may reach past
// its end when HW % 8 != 0: the surplus lands in dead slots (never stored); the caller guarantees the bytes are readable.
__device__ __forceinline__ void offs_mn_conv(unsigned (&off)[4], int hw, int hwp, unsigned hwp_m, long chw, int col0, int cols, int w,
                                             int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = w * 4 + i;
        const int kr = piece * 2 + (lane >> 5);
        const int c_log = (lane & 31) ^ (mn_f(kr) << 1);
        int gc = col0 + c_log * 8;
        gc = gc <= cols - 8 ? gc : cols - 8;
        unsigned img, pix;
        udivmod_m((unsigned)gc, (unsigned)hwp, hwp_m, img, pix);
        off[i] = (unsigned)(((long)img * chw + (long)kr * hw + pix) * 2);
    }
}
__device__ __forceinline__ void stage4(const char *ubase, const unsigned (&off)[4], char *lds_oper, int w) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds(IROCM_GLB_PTR(ubase + (unsigned long)off[i]),
                                         IROCM_LDS_PTR(lds_oper + (w * 4 + i) * 1024), 16, 0, 0);
}

} // namespace g256
} // namespace irocm
