/*
 * infini_rocm.h — C ABI of the MI355X (gfx950 / CDNA4) operator backend for InfiniTensor.
 *
 * This is the drop-in boundary of the hot path `RuntimeObj::run -> Kernel::compute`
 * (reference: include/core/runtime.h:38-101, include/core/kernel.h:32-103). The reference has no
 * C ABI: a backend is a `RuntimeObj` subclass plus `Kernel` subclasses that pull raw device
 * pointers, shapes and attributes out of an operator object and hand them to device code
 * (e.g. src/kernels/cuda/matmul.cc:67-174 -> cuBLAS, src/kernels/cuda/softmax.cc:9-31 ->
 * softmax.cu). Every entry point below is exactly that hand-off — plain pointers, sizes and
 * attributes, no C++ or torch types — so that the C++ plugin in infinitensor_amd/plugin/
 * (`RocmRuntimeObj` + `REGISTER_KERNEL(Device::ROCM, ...)`) and the ctypes host mirror in
 * infinitensor_amd/ bind the same functions. Each declaration cites the reference interface it
 * replaces.
 *
 * Conventions
 *  - All functions return 0 (INFINI_ROCM_OK) on success or a non-zero infiniRocmStatus_t; a
 *    human-readable message for the calling thread is available from infini_rocm_last_error().
 *    The C++ plugin turns a non-zero status into `infini::Exception` (reference: IT_ASSERT,
 *    include/core/common.h:44-55), the Python mirror into RuntimeError.
 *  - dtype codes are the reference's `DataType::getIndex()` values (include/core/data_type.h:8-23).
 *  - All tensors are dense, contiguous, row-major device buffers (reference: src/core/tensor.cc:74-82);
 *    strides passed explicitly are in ELEMENTS and may be 0 for broadcast dimensions.
 *  - Kernels are enqueued asynchronously on the runtime's stream (reference: thread-local
 *    CUDAStream, include/cuda/cuda_common.h:115-139); nothing synchronises unless stated.
 *  - Kernels never allocate persistent device memory; scratch comes from the runtime workspace
 *    (reference: CudaRuntimeObj::getWorkspace, include/cuda/cuda_runtime.h:85-88).
 */
#ifndef INFINI_ROCM_H
#define INFINI_ROCM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    INFINI_ROCM_OK = 0,
    INFINI_ROCM_INVALID_ARGUMENT = 1,
    INFINI_ROCM_UNSUPPORTED = 2,
    INFINI_ROCM_HIP_ERROR = 3,
    INFINI_ROCM_RCCL_ERROR = 4,
    INFINI_ROCM_OUT_OF_MEMORY = 5,
    INFINI_ROCM_CAPTURE_ERROR = 6
} infiniRocmStatus_t;

/* reference: include/core/data_type.h:8-23 (index of each DataType constant) */
typedef enum {
    INFINI_DT_F32 = 1,
    INFINI_DT_U8 = 2,
    INFINI_DT_I8 = 3,
    INFINI_DT_U16 = 4,
    INFINI_DT_I16 = 5,
    INFINI_DT_I32 = 6,
    INFINI_DT_I64 = 7,
    INFINI_DT_BOOL = 9,
    INFINI_DT_F16 = 10,
    INFINI_DT_F64 = 11,
    INFINI_DT_U32 = 12,
    INFINI_DT_U64 = 13,
    INFINI_DT_BF16 = 16
} infiniRocmDType_t;

#define INFINI_ROCM_MAX_DIMS 8 /* reference: SMALL_ARRAY_SIZE, include/utils/small_array.h:4-15 */

typedef struct infiniRocmRuntime *infiniRocmRuntime_t;
typedef struct infiniRocmGraph *infiniRocmGraph_t;
typedef struct infiniRocmEvent *infiniRocmEvent_t;

/* ------------------------------------------------------------------------------------------ */
/* Errors / identification                                                                     */
/* ------------------------------------------------------------------------------------------ */
const char *infini_rocm_last_error(void);
const char *infini_rocm_version(void);
int infini_rocm_device_count(int *count);

typedef struct {
    char name[64];
    char arch[32];
    int compute_units;
    int clock_mhz;          /* max engine clock */
    int memory_clock_mhz;
    int memory_bus_bits;
    size_t total_memory;
    int wavefront_size;
    int lds_bytes_per_cu;
} infiniRocmDeviceInfo;

/* ------------------------------------------------------------------------------------------ */
/* Runtime: one device + one stream + one workspace                                            */
/* replaces CudaRuntimeObj ctor, dtor, alloc, dealloc, copyBlobFromCPU/ToCPU/InsideRuntime, sync */
/* (reference: src/cuda/cuda_runtime.cc:30-120, 481-493; include/cuda/cuda_runtime.h:60-105)    */
/* ------------------------------------------------------------------------------------------ */
int infini_rocm_runtime_create(int device, infiniRocmRuntime_t *out);
int infini_rocm_runtime_destroy(infiniRocmRuntime_t rt);
int infini_rocm_runtime_device_info(infiniRocmRuntime_t rt, infiniRocmDeviceInfo *info);
/* The native hipStream_t the runtime launches on. */
int infini_rocm_runtime_get_stream(infiniRocmRuntime_t rt, void **stream);
/* Adopt an externally owned stream (e.g. torch's current stream). NULL is the legacy default
 * stream, adopted like any other; infini_rocm_runtime_use_own_stream goes back to the own one. */
int infini_rocm_runtime_set_stream(infiniRocmRuntime_t rt, void *stream);
int infini_rocm_runtime_use_own_stream(infiniRocmRuntime_t rt);
int infini_rocm_runtime_sync(infiniRocmRuntime_t rt);
int infini_rocm_alloc(infiniRocmRuntime_t rt, size_t bytes, void **ptr);
int infini_rocm_dealloc(infiniRocmRuntime_t rt, void *ptr);
int infini_rocm_copy_from_cpu(infiniRocmRuntime_t rt, void *dst, const void *src, size_t bytes);
int infini_rocm_copy_to_cpu(infiniRocmRuntime_t rt, void *dst, const void *src, size_t bytes);
/* async device-to-device copy on the runtime stream; also the Reshape/Flatten/Identity/Squeeze/
 * Unsqueeze kernel (reference: src/kernels/cuda/reshape.cc:4-21) */
int infini_rocm_copy_inside(infiniRocmRuntime_t rt, void *dst, const void *src, size_t bytes);
int infini_rocm_memset(infiniRocmRuntime_t rt, void *dst, int value, size_t bytes);
/* Scratch valid until the next call on the same runtime that asks for workspace
 * (reference: getWorkspace, include/cuda/cuda_runtime.h:85-88; grows on demand instead of 7 GiB). */
int infini_rocm_workspace(infiniRocmRuntime_t rt, size_t bytes, void **ptr);
/* Growing the workspace never frees the block it outgrows (hipGraph execs captured earlier still address it; the
 * reference's fixed 7 GiB block never moves either): outgrown blocks are retired and stay allocated until
 * _trim (call it only when no captured graph of this runtime is alive) or runtime destruction. Growth is also
 * legal while the stream is capturing. _info reports the current size, the number of retired blocks and an epoch
 * that changes whenever the current block does. */
int infini_rocm_workspace_trim(infiniRocmRuntime_t rt);
int infini_rocm_workspace_info(infiniRocmRuntime_t rt, size_t *bytes, size_t *retired_blocks, uint64_t *epoch);

/* Diagnostics: one launch of the MFMA-only ceiling kernel (csrc/probe.hip) — the headline GEMM's MFMA stream (8 waves
 * per workgroup, one workgroup per CU, v_mfma_f32_16x16x32 on 8 x 4 accumulator tiles) with no memory or LDS
 * instruction in the loop, on the caller's operand data (>= 1.5 MiB of 16-bit values; sink >= compute_units * 512
 * floats). *flop = FLOP of the launch. Timed with events it gives the matrix-pipe throughput the chip SUSTAINS under
 * its power budget — the attainable denominator next to the nominal 2.5 PFLOP/s (bench.py roofline.attainable_peak). */
int infini_rocm_probe_mfma_ceiling(infiniRocmRuntime_t rt, int dtype, const void *data, void *sink, int iters,
                                   double *flop);
/* The same probe on v_mfma_f32_32x32x16 (4 x 2 accumulator tiles of 32 x 32; equal FLOP per wave and iteration): which instruction
 * shape the chip sustains more of under its power budget (round 5). */
int infini_rocm_probe_mfma_ceiling32(infiniRocmRuntime_t rt, int dtype, const void *data, void *sink, int iters,
                                     double *flop);
/* Round 6: the same MFMA stream with the A fragments of every K-tile loaded straight from an L2-resident [2048][k] panel (16
 * fragment-shaped 16-byte loads per wave and K-tile, register double-buffered) and NO LDS, barrier or B traffic — the upper bound of a
 * GEMM design that streams one operand L2 -> VGPR (the reference has no counterpart: cuBLAS picks its own kernels, matmul.cc:67-174). */
int infini_rocm_probe_mfma_a_from_l2(infiniRocmRuntime_t rt, int dtype, const void *a, const void *bdata, void *sink, int k, int iters,
                                     double *flop);
/* Round 6: four waves per workgroup with 128 x 128 wave tiles (one wave per SIMD, 256 accumulator registers): per K-tile of 64 and wave 128
 * MFMAs, 32 LDS fragment reads and (pieces != 0) 16 LDS-DMA pieces interleaved, one barrier — the upper bound of that tile shape with the
 * library's staging machinery; timing only (no counterpart in the reference: matmul.cc:67-174 calls cuBLAS). pieces: 0 none, 1 with a
 * full wait per K-tile, 2 one K-tile's pieces left in flight, 3 pieces but no fragment reads, 4 classic staging (global_load_dwordx4 +
 * ds_write_b128), 5 mode 2 with the waves skewed, 7 mode 2 into LDS nobody reads (random against zero source data), 8 / 9 the SPREAD
 * schedule (every non-MFMA instruction alone between two MFMAs, four-stage ring: what csrc/gemm128w.hip implements) with / without its
 * pieces. Workgroup 0 leaves (core-clock ticks, 100 MHz ticks) of its loop in the first 16 bytes of sink. */
int infini_rocm_probe_mfma_wave128(infiniRocmRuntime_t rt, int dtype, const void *panel, void *sink, int pieces, int iters, double *flop);
/* Diagnostics: one launch of the persistent GEMM (bf16, row-major A [m,k] and B [k,n], no bias; tile_cols 256 or 192)
 * built with s_memtime stamps at every wave's phase boundaries. trace: [min(tiles, compute_units)][8][128] uint64
 * (0 = unused): slot 0 kernel entry, then per K-tile {L1 start, L2 start}, per tile {epilogue start, end}, last = after
 * the final store drain. tools/gemm_timeline.py turns it into a per-phase table. */
int infini_rocm_probe_gemm_timeline(infiniRocmRuntime_t rt, const void *a, const void *b, void *c, int64_t m, int64_t n,
                                    int64_t k, int tile_cols, void *trace);

/* Events on the runtime stream, for timing (reference: timeit, src/core/common.cc:7-22). */
int infini_rocm_event_create(infiniRocmEvent_t *ev);
int infini_rocm_event_destroy(infiniRocmEvent_t ev);
int infini_rocm_event_record(infiniRocmRuntime_t rt, infiniRocmEvent_t ev);
int infini_rocm_event_elapsed_ms(infiniRocmEvent_t start, infiniRocmEvent_t stop, float *ms);

/* hipGraph capture/replay of whatever is enqueued on the runtime stream between begin and end
 * (reference: runWithCudaGraph / captureGraph, src/cuda/cuda_runtime.cc:252-426). */
int infini_rocm_graph_begin_capture(infiniRocmRuntime_t rt);
int infini_rocm_graph_end_capture(infiniRocmRuntime_t rt, infiniRocmGraph_t *graph);
int infini_rocm_graph_abort_capture(infiniRocmRuntime_t rt);
int infini_rocm_graph_launch(infiniRocmRuntime_t rt, infiniRocmGraph_t graph);
int infini_rocm_graph_destroy(infiniRocmGraph_t graph);

/* ------------------------------------------------------------------------------------------ */
/* MatMul  (reference: matmulCublas::do_compute, src/kernels/cuda/matmul.cc:67-174;             */
/*          op definition src/operators/matmul.cc:26-49)                                        */
/*   C[b,m,n] = op(A)[b,m,k] . op(B)[b,k,n] (+ bias), row-major.                                */
/*   A is stored [m,k] (trans_a=0) or [k,m] (trans_a=1); B is [k,n] or [n,k] (trans_b=1).       */
/*   stride_a / stride_b: batch stride in elements, 0 = the operand is broadcast over the batch */
/*   (reference matmul.cc:124-137). bias (may be NULL) is addressed as                          */
/*   bias[ib*bias_stride_b + im*bias_stride_m + in*bias_stride_n] (reference expands it into C  */
/*   and runs the GEMM with beta=1, matmul.cc:86-118).                                          */
/*   dtype F32: exact-f32 MFMA (v_mfma_f32_32x32x2_f32), fp32 accumulate.                       */
/*   dtype F16 / BF16: v_mfma_f32_16x16x32_{f16,bf16}, fp32 accumulate, one rounding on store.  */
/*   act: 0 none, 1 relu, 2 sigmoid, 3 tanh (reference ActType, include/core/common.h), 4 gelu  */
/*   (erf form, erff), 5 gelu for 16-bit outputs: erf by a clamped odd polynomial, no            */
/*   transcendentals, ABSOLUTE error < 2^-12 — NOT a relative bound in the negative tail, see   */
/*   INFINI_UN_GELU (what the runtime's MatMul -> Gelu fusion uses for f16 / bf16 outputs; an   */
/*   fp32 caller wants 4); the                                                                  */
/*   reference CUDA kernel ignores it — the plugin passes 0 to stay op-for-op identical).       */
/* ------------------------------------------------------------------------------------------ */
int infini_rocm_matmul(infiniRocmRuntime_t rt, int dtype, const void *a, const void *b,
                       const void *bias, void *c, int64_t batch, int64_t m, int64_t n, int64_t k,
                       int trans_a, int trans_b, int64_t stride_a, int64_t stride_b,
                       int64_t bias_stride_b, int64_t bias_stride_m, int64_t bias_stride_n,
                       int act);
/* MatmulObj::getComputeType() (reference: matmul.cc:51-64): how an fp32 MatMul forms its PRODUCTS. 0 = "default" and "tf32":
 * exact fp32 (gfx950 has no xf32 matrix instruction; more accurate than asked). 1 = "bf16", 2 = "fp16": A and B are converted
 * once (workspace) and multiplied on the 16-bit MFMA path with fp32 accumulation and fp32 output (~10x the exact kernel's
 * rate) wherever that kernel serves the shape (K % 64 == 0, 16-byte aligned rows), else exactly. Sticky per runtime until
 * reset to 0; 16-bit MatMuls ignore it. */
int infini_rocm_matmul_set_compute_type(infiniRocmRuntime_t rt, int compute_type);
/* MatMul whose result is stored head-split: the [m x n] block of every batch entry is written as
 * [m / seq][n / head_dim][seq][head_dim], i.e. MatMul -> Reshape([B, S, H, D]) -> Transpose(0, 2, 1, 3) — the q / k / v
 * projection of a transformer layer as ONNX exporters emit it (reference: matmul.cc + CopyCuda reshape.cc:4-13 +
 * TransposeCuda transpose.cc:8-45, three launches) — in the GEMM's own epilogue. seq = head_dim = 0: plain MatMul.
 * head_dim % 8 == 0, seq | m, head_dim | n. Same sums and rounding as infini_rocm_matmul. */
int infini_rocm_matmul_headsplit(infiniRocmRuntime_t rt, int dtype, const void *a, const void *b,
                                 const void *bias, void *c, int64_t batch, int64_t m, int64_t n, int64_t k,
                                 int trans_a, int trans_b, int64_t stride_a, int64_t stride_b,
                                 int64_t bias_stride_b, int64_t bias_stride_m, int64_t bias_stride_n,
                                 int act, int64_t seq, int64_t head_dim);
/* The same with an explicit element stride between the batches' [m x n] output blocks (stride_c; 0 = m * n, i.e. exactly
 * infini_rocm_matmul_headsplit). Replaces several matmulCublas calls (src/kernels/cuda/matmul.cc:67-174, one per MatMul
 * operator) that share their left operand: lets a caller run several MatMuls of ONE left operand that write separate tensors as one
 * launch — batch index = member, stride_a = 0, stride_b / bias_stride_b / stride_c = the (uniform) distances between the
 * members' weights / biases / outputs (rocm_fusion.cc: gate + up, q + k + v of a decoder block). */
int infini_rocm_matmul_grouped(infiniRocmRuntime_t rt, int dtype, const void *a, const void *b,
                               const void *bias, void *c, int64_t batch, int64_t m, int64_t n, int64_t k,
                               int trans_a, int trans_b, int64_t stride_a, int64_t stride_b, int64_t stride_c,
                               int64_t bias_stride_b, int64_t bias_stride_m, int64_t bias_stride_n,
                               int act, int64_t seq, int64_t head_dim);
/* May a matmul of this extent take the runtime workspace (split-K partial planes) under the current variant setting?
 * A caller that parks an operand of the NEXT matmul in the workspace (rocm_fusion.cc: the fused attention's output
 * feeding the output projection) asks first. *may = 1 is conservative: split-K is possible, not certain. */
int infini_rocm_matmul_may_use_workspace(infiniRocmRuntime_t rt, int64_t batch, int64_t m, int64_t n, int *may);
/* Select a specific GEMM kernel variant for the next matmul calls on this runtime
 * (-1 = heuristic). Used by tune() (reference: 24-algo sweep, matmul.cc:187-208) and by bench.py. */
int infini_rocm_matmul_set_variant(infiniRocmRuntime_t rt, int variant);
int infini_rocm_matmul_num_variants(void);
const char *infini_rocm_matmul_variant_name(int variant);
/* The variant the most recent matmul call on this runtime actually launched (-1 before the first): what the heuristic /
 * the forced setting resolved to for that shape. Measurement tools stamp their records with it (bench.py refuses counter
 * files taken from another kernel). */
int infini_rocm_matmul_last_variant(infiniRocmRuntime_t rt, int *variant);

/* ------------------------------------------------------------------------------------------ */
/* Softmax along one axis (reference: softmax_kernel, src/kernels/cuda/softmax.cu:242-404;      */
/*   glue src/kernels/cuda/softmax.cc:9-31). The tensor is viewed as [outer, dimsize, inner]    */
/*   (inner = stride of `axis`), F32 / F16 / BF16, fp32 math.                                   */
/* ------------------------------------------------------------------------------------------ */
int infini_rocm_softmax(infiniRocmRuntime_t rt, int dtype, const void *x, void *y, int64_t outer,
                        int64_t dimsize, int64_t inner);

/* ------------------------------------------------------------------------------------------ */
/* LayerNormalization (reference: LaynormKernel, src/kernels/cuda/layer_norm.cu:339-558;        */
/*   glue src/kernels/cuda/layer_norm.cc:9-58). x viewed as [outer, norm_size] with norm_size = */
/*   prod(dims[axis..]) (ONNX-17 semantics; identical to the reference whenever axis is the     */
/*   last dim, which is all its tests cover). scale/bias have scale_size/bias_size elements:    */
/*   either norm_size (element j uses [j]) or 1 (scalar broadcast) (layer_norm.cu:41-89).       */
/*   bias may be NULL. Statistics always accumulate in fp32 (deliberate deviation: the          */
/*   reference fp16 path accumulates in half, layer_norm.cu:12,26).                             */
/* ------------------------------------------------------------------------------------------ */
int infini_rocm_layer_norm(infiniRocmRuntime_t rt, int dtype, const void *x, const void *scale,
                           const void *bias, void *y, int64_t outer, int64_t norm_size,
                           int64_t scale_size, int64_t bias_size, float eps);

/* RMSNorm: y = x * rsqrt(mean(x^2) + eps) * w over the last dim
 * (reference: src/kernels/cuda/rms_norm.cu:35-54; eps is hard-coded 1e-5 there, rms_norm.cu:46). */
int infini_rocm_rms_norm(infiniRocmRuntime_t rt, int dtype, const void *x, const void *w, void *y,
                         int64_t outer, int64_t norm_size, float eps);
/* y = LayerNorm / RMSNorm (rms != 0) of (a + b): the residual join in front of a transformer normalisation in one
 * pass; the sum is rounded to the storage type first like in Add -> Norm (equal to it up to rounding ties). Arguments as layer_norm. */
int infini_rocm_add_norm(infiniRocmRuntime_t rt, int dtype, int rms, const void *a, const void *b, const void *scale,
                         const void *bias, void *y, int64_t outer, int64_t norm_size, int64_t scale_size,
                         int64_t bias_size, float eps);
/* y = LayerNorm / RMSNorm of ((a + pre) + b) with `pre` ONE row of norm_size elements (NULL: add_norm): the chain
 * MatMul -> Add(bias) -> Add(residual) -> LayerNormalization the reference's ONNX front-end emits for a transformer's
 * output projections (pyinfinitensor/onnx.py:280-290 imports MatMul without bias; element_wise.cc, layer_norm.cc) when the
 * bias could not ride in the GEMM epilogue. Both sums are rounded to the storage type like their own Add kernels. */
int infini_rocm_bias_add_norm(infiniRocmRuntime_t rt, int dtype, int rms, const void *a, const void *pre, const void *b,
                              const void *scale, const void *bias, void *y, int64_t outer, int64_t norm_size,
                              int64_t scale_size, int64_t bias_size, float eps);

/* Fused prefill attention  O = softmax(scale * Q K^T + mask) V  per (batch x head); f16 / bf16, head dim 64 or 128.
 * Replaces the chain MatMul(Q, K^T) -> Div/Mul(scalar) -> Add(mask) -> Softmax -> MatMul(P, V) of the reference graph
 * (matmul.cc:67-174, element_wise.cu, softmax.cu) without materialising the score matrix.
 *   q, out: [batch_heads, seq_q, head_dim]; k, v: [batch_heads, seq_k, head_dim]; dense, 16-byte aligned.
 *   mask (optional, same dtype): additive, [batch_heads / mask_group, seq_k], broadcast over the query rows
 *     (a BERT [B,1,1,S] padding mask has mask_group = heads; [1,1,1,S] has mask_group = batch_heads).
 *   scale: scale_dev != NULL -> one element of `dtype` in device memory, applied as multiply (scale_is_div = 0) or
 *     divide (1); otherwise the immediate `scale`.
 *   causal: keys with index > query index + (seq_k - seq_q) are excluded. Rows with no admissible key give 0. */
int infini_rocm_attention(infiniRocmRuntime_t rt, int dtype, const void *q, const void *k, const void *v,
                          const void *mask, void *out, int64_t batch_heads, int64_t seq_q, int64_t seq_k,
                          int64_t head_dim, int64_t mask_group, const void *scale_dev, int scale_is_div,
                          float scale, int causal);

/* The same attention with two extensions used by the runtime's fusion:
 *   heads > 0: the result is stored head-merged, out = [batch_heads / heads, seq_q, heads, head_dim], i.e. the
 *     Transpose(0, 2, 1, 3) -> Reshape([B, S, H * D]) a transformer layer applies to the context before its output
 *     projection (reference: TransposeCuda transpose.cc:8-45 + CopyCuda reshape.cc:4-13) done by the kernel's own store;
 *   mask_2d != 0: mask is a full additive mask [batch_heads / mask_group, seq_q, seq_k] (one row per query: a causal or
 *     sliding-window mask passed as a tensor, as exported decoder graphs do) instead of one row per key sequence.
 * heads = 0, mask_2d = 0: exactly infini_rocm_attention. */
int infini_rocm_attention_ex(infiniRocmRuntime_t rt, int dtype, const void *q, const void *k, const void *v,
                             const void *mask, void *out, int64_t batch_heads, int64_t seq_q, int64_t seq_k,
                             int64_t head_dim, int64_t mask_group, const void *scale_dev, int scale_is_div,
                             float scale, int causal, int64_t heads, int mask_2d);

/* AttentionKVCache: one decode step with in-place cache append (reference: attention_kvcache.cu:8-169).
 *   n = position_id[0] + 1; k_cache/v_cache[bh, n-1, :] = k/v[bh, :]; out[bh, :] = softmax(q . K[0:n]^T / sqrt(D)) V[0:n].
 * caches [batch_heads, max_seq, D]; q, k, v, out [batch_heads, D]; position_id: device I32 / U32 / I64 (element 0 is
 * used for all heads, as in the reference). f32 (the reference's only type) / f16 / bf16; D in {128, 256}.
 * Round 5: the cache is cut into G chunks over workgroups (~one workgroup per CU; n is read on the device, the chunk length is
 * computed in the kernel) and a second small kernel merges the G fp32 partials (m, l, o[D]) per head — the reference splits over
 * gridDim.y = ceil(S / 16) and merges the same way (attention_kvcache.cu:18-25, 118-166); G x batch_heads x (D + 2) floats come from
 * the runtime workspace. 16-byte aligned tensors; others keep the one-workgroup element-wise kernel. */
int infini_rocm_attention_kvcache(infiniRocmRuntime_t rt, int dtype, void *k_cache, void *v_cache, const void *q,
                                  const void *k, const void *v, int pos_dtype, const void *position_id, void *out,
                                  int64_t batch_heads, int64_t max_seq, int64_t head_dim);

/* RoPE, rotate-half form (reference: _rope_kernel, src/kernels/cuda/rope.cu:6-31; glue rope.cc:8-33).
 * x, y: [tokens, dim_model]; a trailing partial head (dim_model % dim_head != 0) is accepted and its missing
 * partner columns count as 0 (what test_cuda_rope.cc:17-31 relies on: dim_model 32, head dim 128); pos: one
 * position per token (I32 / U32 / I64). The reference hard-codes dim_head = 128 and theta = 10000 and its launch covers a
 * single (batch, position) (rope.cu:85): here every token is rotated. */
int infini_rocm_rope(infiniRocmRuntime_t rt, int dtype, int pos_dtype, const void *pos, const void *x,
                     void *y, int64_t tokens, int64_t dim_model, int64_t dim_head, float theta);
/* RoPE with a head-split store: token (b, s), head h, column c goes to y[b][h][s][c] — RoPE -> Reshape([B, S, H, D]) ->
 * Transpose(0, 2, 1, 3) (rope.cu, reshape.cc, transpose.cc: three launches) as one pass. seq = S (0: exactly infini_rocm_rope);
 * tokens % seq == 0, dim_model % dim_head == 0, x != y. */
int infini_rocm_rope_headsplit(infiniRocmRuntime_t rt, int dtype, int pos_dtype, const void *pos, const void *x,
                               void *y, int64_t tokens, int64_t dim_model, int64_t dim_head, float theta, int64_t seq);

/* ------------------------------------------------------------------------------------------ */
/* Binary element-wise with numpy broadcasting                                                  */
/* (reference: ElementWiseCudnn / ElementWiseCuda, src/kernels/cuda/element_wise.cc:13-175,     */
/*  element_wise.cu:9-131; semantics src/kernels/cpu/element_wise.cc:43-112).                   */
/*   c[i] = a[bcast(i)] OP b[bcast(i)]; ndim <= 8; shape = output shape;                        */
/*   stride_a/stride_b in elements over the OUTPUT index space (0 where broadcast).             */
/*   Comparison ops write 1/0 in the input dtype (reference CPU kernel: `(T)(a < b)`).          */
/* ------------------------------------------------------------------------------------------ */
typedef enum {
    INFINI_BIN_ADD = 0,
    INFINI_BIN_SUB = 1,
    INFINI_BIN_MUL = 2,
    INFINI_BIN_DIV = 3,
    INFINI_BIN_POW = 4,
    INFINI_BIN_MIN = 5,
    INFINI_BIN_MAX = 6,
    INFINI_BIN_EQUAL = 7,
    INFINI_BIN_GREATER = 8,
    INFINI_BIN_GREATER_EQUAL = 9,
    INFINI_BIN_LESS = 10,
    INFINI_BIN_LESS_EQUAL = 11,
    INFINI_BIN_ADD_RELU = 12 /* max(a + b, 0): the fused residual join (Add -> Relu), bit-identical to the chain */
} infiniRocmBinaryOp_t;

int infini_rocm_binary(infiniRocmRuntime_t rt, int op, int dtype, const void *a, const void *b,
                       void *c, int ndim, const int64_t *shape, const int64_t *stride_a,
                       const int64_t *stride_b);
/* out[o, c, i] = act(a[o, c, i] + bias[c] + residual[o, c, i]) with the intermediate sum rounded like the unfused
 * Add -> Add [-> Relu] chain (bit-identical to it); relu: 0 / 1. One pass instead of two or three. */
int infini_rocm_bias_residual(infiniRocmRuntime_t rt, int dtype, const void *a, const void *bias, const void *residual,
                              void *out, int64_t outer, int64_t channels, int64_t inner, int relu);

/* ------------------------------------------------------------------------------------------ */
/* Unary element-wise (reference: unary_kernel, src/kernels/cuda/unary.cu:262-352; cuDNN        */
/*  activations src/kernels/cuda/unary.cc:70-122; formulas src/kernels/cpu/unary.cc:8-72).      */
/*  p0/p1: Clip min/max (NaN = absent), Elu/LeakyRelu alpha, HardSigmoid alpha/beta.            */
/* ------------------------------------------------------------------------------------------ */
typedef enum {
    INFINI_UN_RELU = 0,
    INFINI_UN_SIGMOID = 1,
    INFINI_UN_TANH = 2,
    INFINI_UN_ABS = 3,
    INFINI_UN_SQRT = 4,
    INFINI_UN_GELU = 5,       /* 0.5 x (1 + erf(x / sqrt 2)). fp32: erff. f16 / bf16: erf by the clamped odd polynomial the GEMM epilogue
                               * uses (act 5 of infini_rocm_matmul) so that MatMul -> Gelu is bit-identical fused or not: ABSOLUTE error
                               * < 2^-12 (1.96e-4) everywhere — below half an ulp of the stored 16-bit value for |gelu(x)| >= 0.25, but
                               * a growing RELATIVE error in the negative tail, where gelu(x) itself shrinks towards that bound: <= 0.3 % on
                               * [-2, -1], <= 2.5 % on [-3, -2], <= 12 % on [-3.5, -3], of the order of the value itself below -3.5 (true
                               * value -8e-4 .. -1.3e-4) and exactly 0 from x <= -4 on. A caller that needs the tail takes the fp32 form. */
    INFINI_UN_SILU = 6,
    INFINI_UN_NEG = 7,
    INFINI_UN_ERF = 8,
    INFINI_UN_HARD_SIGMOID = 9, /* max(0, min(1, 0.2 x + 0.5)) */
    INFINI_UN_HARD_SWISH = 10,  /* x * max(0, min(1, x/6 + 0.5)) */
    INFINI_UN_EXP = 11,
    INFINI_UN_LOG = 12,
    INFINI_UN_RECIPROCAL = 13,
    INFINI_UN_ELU = 14,
    INFINI_UN_LEAKY_RELU = 15,
    INFINI_UN_CLIP = 16,
    INFINI_UN_SIN = 17,
    INFINI_UN_COS = 18,
    INFINI_UN_CEIL = 19,
    INFINI_UN_FLOOR = 20,
    INFINI_UN_ROUND = 21
} infiniRocmUnaryOp_t;

int infini_rocm_unary(infiniRocmRuntime_t rt, int op, int dtype, const void *x, void *y,
                      int64_t n, float p0, float p1);

/* y = silu(a) * b over n elements, same shape (a gated MLP's Silu -> Mul pair, reference: unary.cu silu + element_wise.cu mul,
 * two launches and one extra pass over the tensor). The Silu value is rounded to the tensor's dtype before the product:
 * bit-identical to the two-kernel chain. Operands 16-byte aligned; y may alias a or b. */
int infini_rocm_silu_mul(infiniRocmRuntime_t rt, int dtype, const void *a, const void *b, void *y, int64_t n);

/* Cast between dtypes (reference: CastCuda, src/kernels/cuda/unary.cc:30-68, 5 of the 26
 * CastTypes; here any pair of {F32,F16,BF16,F64,I8,U8,I16,I32,I64,U32,BOOL}). Float->int
 * truncates toward zero like a C cast (reference cast kernel: `(T)x`). */
int infini_rocm_cast(infiniRocmRuntime_t rt, int src_dtype, int dst_dtype, const void *x, void *y,
                     int64_t n);

/* ------------------------------------------------------------------------------------------ */
/* Conv2d (reference: convCudnn, src/kernels/cuda/conv.cc:57-168; op src/operators/conv.cc:47-114; */
/*   index math src/kernels/cpu/conv.cc:25-50). x [n,c,h,w], w [f, c/groups, r, s] -> y [n,f,oh,ow] */
/*   with oh = (h - (r - sh)*dh + 2*ph) / sh (conv.cc:98-101). Cross-correlation, symmetric zero     */
/*   padding. F16/BF16: implicit-GEMM on MFMA (fp32 accumulate); F32: exact fp32 fma chain.          */
/*   bias ([f], may be NULL) and act (0 none, 1 relu, 2 sigmoid, 3 tanh) allow fusing the            */
/*   Conv -> Add(bias) -> Relu chain the front-end emits (onnx.py:159-190); the reference Conv op    */
/*   itself has neither (conv.cc:68-69, conv.cc:143-168), and the plugin passes NULL / 0.            */
/* ------------------------------------------------------------------------------------------ */
int infini_rocm_conv2d(infiniRocmRuntime_t rt, int dtype, const void *x, const void *w,
                       const void *bias, void *y, int64_t n, int64_t c, int64_t h, int64_t wd,
                       int64_t f, int64_t r, int64_t s, int ph, int pw, int sh, int sw, int dh, int dw,
                       int64_t groups, int act);
/* conv2d with a residual: y = act(conv(x, w) + bias + residual), residual of y's shape (optional). Used by the
 * runtime's fusion of Conv -> Add(bias) -> Add(identity) -> Relu (the tail of every ResNet bottleneck). */
int infini_rocm_conv2d_res(infiniRocmRuntime_t rt, int dtype, const void *x, const void *w, const void *bias,
                           const void *residual, void *y, int64_t n, int64_t c, int64_t h, int64_t wd, int64_t f, int64_t r,
                           int64_t s, int ph, int pw, int sh, int sw, int dh, int dw, int64_t groups, int act);
/* ConvTranspose2d (reference: convBackwardDataCudnn, src/kernels/cuda/conv_transposed.cc:46-230; shape rule
 * src/operators/conv.cc:252-268): x [n, f, h, w], w [f, c_per_group, r, s] -> y [n, c_per_group * groups, oh, ow],
 * oh = (h - 1) sh - 2 ph + dh (r - 1) + oph + 1. bias ([C], optional) and act as for conv2d. Gather-form direct
 * kernel, fp32 accumulation. */
int infini_rocm_conv_transpose2d(infiniRocmRuntime_t rt, int dtype, const void *x, const void *w, const void *bias,
                                 void *y, int64_t n, int64_t f, int64_t h, int64_t wd, int64_t c_per_group, int64_t r,
                                 int64_t s, int ph, int pw, int sh, int sw, int dh, int dw, int oph, int opw,
                                 int64_t groups, int act);
/* Kernel choice for the next f16 / bf16 conv2d calls: -1 heuristic, 1 generic implicit GEMM only,
 * 2 the conv_s1.hip kernels (LDS-resident input patch for unit-stride "same" R x S, tap-shifted implicit GEMM
 * otherwise) for every shape they serve, 3 batched-GEMM route for every eligible pointwise shape, 4 = 2 with the
 * patch kernel off (the tap-shifted kernel everywhere: A/B), 5 = pointwise layers (1 x 1, any stride, channels % 64 == 0) as ONE
 * GEMM over pixel slots on the persistent 256-row kernels for every shape that qualifies (the heuristic sends them there when
 * they have >= 128 filters and enough tiles to fill half the chip), 6 = 2 with the 8-wave 128 x 256 form of the patch kernel for every
 * shape it serves (the heuristic picks it when its workgroups still fill most of the chip), 7 = 3 x 3 / pad 1 layers of stride 1 or 2
 * (channels % 64 == 0, no residual) as ONE GEMM with K = 9 C over pixel slots on the persistent 256-row kernels ("tap mode": a tap
 * moves the pointwise tile's 16-byte runs, the zero padding is a per-lane mask on the B fragments; the heuristic sends layers with
 * >= 256 filters and enough slot tiles there). All variants compute the same sums (fp32 accumulate).
 * Used by tune() and tests. */
int infini_rocm_conv2d_set_variant(infiniRocmRuntime_t rt, int variant);
/* Which implementation the most recent conv2d call on this runtime launched: "igemm32" / "batched_gemm32" (fp32 on the fp32 matrix
 * instruction: implicit GEMM, or a unit-stride pointwise layer as a batched GEMM), "direct32" (fp32, one output per thread: grouped layers,
 * C R S % 4 != 0, conv variant 1), "pixel_gemm" (pointwise layer as
 * one GEMM over pixel slots on the persistent kernels), "tap_gemm" (3 x 3 layer as one GEMM with K = 9 C on the same kernels), "depthwise" (groups == C, 3 x 3 / 5 x 5, stride 1 / 2: the HBM-bound kernel of conv_dw.hip), "resident" (F <= 64, C <= 64: weights resident in LDS, persistent
 * workgroups), "tap_shifted" (the other kernels of conv_s1.hip), "batched_gemm", "generic", "none". A forced
 * variant falls back when a shape does not qualify; tests and measurement tools read the route instead of assuming it. */
int infini_rocm_conv2d_last_route(infiniRocmRuntime_t rt, const char **route);
/* The stem of a CNN as one launch (csrc/conv_stem.hip): y = MaxPool(pool_k x pool_k, stride pool_s, pad pool_p)(act(conv2d(x, w) +
 * bias)) — the chain Conv -> Reshape(bias) -> Add -> Relu -> MaxPool the front-end emits for ResNet's first layers (reference:
 * conv.cc:57-168, element_wise.cu, unary.cc:70-122, pooling.cc:6-95 as four kernels). The conv tile is pooled out of LDS: the conv
 * output never exists in HBM. Served: 7 x 7 / stride 2 / pad 3, C = 3, F = 64, groups 1, act = 1 (ReLU), MaxPool 3 x 3 / 2 / 1,
 * f16 / bf16, W % 8 == 0, 16-byte aligned x / y; infini_rocm_conv2d_pool_supported says so without a runtime (the plugin's planner
 * asks it); anything else returns INFINI_ROCM_UNSUPPORTED — the caller runs the separate kernels. y: [n, f, ph, pw] with
 * oh = (h + 2 ph - r) / sh + 1 and ph = (oh + 2 pool_p - pool_k) / pool_s + 1 (floor mode). */
int infini_rocm_conv2d_pool_supported(int dtype, int64_t c, int64_t h, int64_t w, int64_t f, int64_t r, int64_t s, int ph, int pw, int sh,
                                      int sw, int dh, int dw, int groups, int act, int pool_k, int pool_s, int pool_p);
int infini_rocm_conv2d_pool(infiniRocmRuntime_t rt, int dtype, const void *x, const void *w, const void *bias, void *y, int64_t n, int64_t c,
                            int64_t h, int64_t wd, int64_t f, int64_t r, int64_t s, int ph, int pw, int sh, int sw, int dh, int dw, int groups,
                            int act, int pool_k, int pool_s, int pool_p);
/* Packed-weight cache. The f16 / bf16 conv kernels read their weights re-packed (FCRS -> [RS][F][C]); while `on` is set
 * the next conv2d calls treat `w` as CONSTANT data: the packed image is built once, kept in a runtime-owned buffer keyed
 * by (w, F, C, RS, layout) and reused by every later call — eager or captured (built on a side stream when the runtime
 * stream is recording, so a hipGraph contains no pack node). An image is dropped when copy_from_cpu / copy_inside /
 * memset / dealloc touch its source range (weight_cache_info's epoch then changes: whoever holds captured graphs of this
 * runtime must drop them — the plugin does) or by weight_cache_clear. Off (default): re-packed per call in the workspace.
 * The reference's cuDNN path has no per-call weight transform either (src/kernels/cuda/conv.cc:143-168). */
int infini_rocm_conv2d_set_const_weights(infiniRocmRuntime_t rt, int on);
int infini_rocm_weight_cache_info(infiniRocmRuntime_t rt, size_t *entries, size_t *bytes, uint64_t *epoch);
int infini_rocm_weight_cache_clear(infiniRocmRuntime_t rt);

/* ------------------------------------------------------------------------------------------ */
/* ReduceSum / ReduceMean over arbitrary axes (reference: ReduceCudnnBase::compute,             */
/*   src/kernels/cuda/reduce.cc:10-108). kind: 0 sum, 1 mean. reduced[d] != 0 marks reduced dims; */
/*   the output is dense over the kept dims (keepdims only changes the shape, not the data).     */
/* ------------------------------------------------------------------------------------------ */
int infini_rocm_reduce(infiniRocmRuntime_t rt, int kind, int dtype, const void *x, void *y, int ndim,
                       const int64_t *shape, const int *reduced);

/* BatchNormalization, inference (reference: BatchNormCudnn, src/kernels/cuda/batch_norm.cc:7-67;   */
/*   operator input order x, mean, var, scale, bias — include/operators/batch_norm.h:10-50).         */
/*   x [n, c, inner]; mean/var/scale/bias are F32 [c] whatever the dtype of x.                        */
int infini_rocm_batch_norm(infiniRocmRuntime_t rt, int dtype, const void *x, const void *mean,
                           const void *var, const void *scale, const void *bias, void *y, int64_t n,
                           int64_t c, int64_t inner, float eps);

/* MaxPool / AveragePool 2-D, NCHW (reference: poolingCudnn, src/kernels/cuda/pooling.cc:6-95;      */
/*   output size src/operators/pooling.cc:17-35). kind: 0 max, 1 average. Average divides by kh*kw   */
/*   (COUNT_INCLUDE_PADDING, pooling.cc:86-90); max ignores padding. Dilation is honoured (ONNX);     */
/* LRN across channels, ONNX semantics (reference operator src/operators/lrn.cc; its only kernel is Cambricon's,
 * src/kernels/bang/lrn.cc:6-56, and no test pins values: the oracle is the ONNX definition). x, y: [n, c, inner];
 * y = x / (bias + alpha / size * sum of x^2 over channels [c - floor((size-1)/2), c + ceil((size-1)/2)])^beta. f32/f16/bf16. */
int infini_rocm_lrn(infiniRocmRuntime_t rt, int dtype, const void *x, void *y, int64_t n, int64_t c, int64_t inner,
                    int size, float alpha, float beta, float bias);

/*   cuDNN ignores it.                                                                               */
int infini_rocm_pool2d(infiniRocmRuntime_t rt, int kind, int dtype, const void *x, void *y, int64_t n,
                       int64_t c, int64_t h, int64_t w, int kh, int kw, int dh, int dw, int ph, int pw,
                       int sh, int sw, int ceil_mode);
/* max pooling of relu(x) (relu != 0; kind must be 0): Relu -> MaxPool in one pass — max and relu commute, so the result
 * is bit-identical to the chain. relu == 0: plain pool2d. */
int infini_rocm_pool2d_relu(infiniRocmRuntime_t rt, int kind, int dtype, const void *x, void *y, int64_t n,
                            int64_t c, int64_t h, int64_t w, int kh, int kw, int dh, int dw, int ph, int pw,
                            int sh, int sw, int ceil_mode, int relu);

/* ------------------------------------------------------------------------------------------ */
/* Data movement / indexing: bit-exact for every dtype (raw 1/2/4/8-byte elements).            */
/* ------------------------------------------------------------------------------------------ */
/* Transpose: y = permute(x, perm), rank <= 8 (reference: TransposeCuda, src/kernels/cuda/transpose.cc:8-45,
 * transpose.cu:10-24). in_shape is the INPUT shape; output dim d has extent in_shape[perm[d]]. */
int infini_rocm_transpose(infiniRocmRuntime_t rt, int dtype, const void *x, void *y, int ndim,
                          const int64_t *in_shape, const int *perm);
/* Expand: broadcast copy (reference: ExpandCuda, src/kernels/cuda/expand.cu:10-49). x_strides are the
 * element strides of x over the OUTPUT index space (0 where broadcast). */
int infini_rocm_expand(infiniRocmRuntime_t rt, int dtype, const void *x, void *y, int ndim,
                       const int64_t *out_shape, const int64_t *x_strides);
/* Gather along one axis (reference: GatherCuda, src/kernels/cuda/gather.cu:4-39, include/cuda/gather.h:33-55).
 * data viewed as [outer, axis_dim, inner]; indices (I32 or I64, n_indices of them, negative wraps) ->
 * y [outer, n_indices, inner]. */
int infini_rocm_gather(infiniRocmRuntime_t rt, int dtype, int index_dtype, const void *data,
                       const void *indices, void *y, int64_t outer, int64_t axis_dim, int64_t n_indices,
                       int64_t inner);
/* Resize (reference: src/kernels/cuda/resize.cu:6-196, resize.cc:5-50). x [in_shape] -> y [out_shape], same rank.
 * scales[d] = the operator's per-dim scale (out / in as ResizeObj computed it, src/operators/resize.cc:75-230);
 * roi: 2*ndim floats (starts, ends) or NULL. mode 0 nearest / 1 linear / 2 cubic (A = -0.75);
 * coord_mode 0 half_pixel / 1 pytorch_half_pixel / 2 align_corners / 3 asymmetric / 4 tf_crop_and_resize;
 * nearest_mode 0 round_prefer_floor / 1 round_prefer_ceil / 2 floor / 3 ceil. f32 / f16 / bf16. */
int infini_rocm_resize(infiniRocmRuntime_t rt, int dtype, const void *x, void *y, int ndim,
                       const int64_t *in_shape, const int64_t *out_shape, const float *scales,
                       const float *roi, int mode, int coord_mode, int nearest_mode);
/* GatherElements (reference: _gather_elements_kernel, src/kernels/cuda/gather_elements.cu:4-35; operator
 * gather_elements.cc:27-39): y has index_shape; y[i] = data[i with coordinate `axis` replaced by indices[i]].
 * indices I32 / I64 (negative wraps); data and index ranks equal, off-axis index extents <= data extents. */
int infini_rocm_gather_elements(infiniRocmRuntime_t rt, int dtype, int index_dtype, const void *data,
                                const void *indices, void *y, int ndim, const int64_t *data_shape,
                                const int64_t *index_shape, int axis);
/* Where: out = cond ? x : y with 3-way broadcast; cond is 1 byte per element (reference: WhereCuda,
 * src/kernels/cuda/where.cu:4-63). Strides are element strides over the OUTPUT index space. */
int infini_rocm_where(infiniRocmRuntime_t rt, int dtype, const void *x, const void *y, const void *cond,
                      void *out, int ndim, const int64_t *shape, const int64_t *stride_x,
                      const int64_t *stride_y, const int64_t *stride_c);
/* Same, with the condition read in `cond_dtype` (non-zero = true; -0.0 is false). The reference's CUDA Less writes
 * bool BYTES into an output the graph declares with the operand dtype (element_wise.cu:101-131) and WhereCuda reads
 * bytes; this backend's comparisons write whole elements (as the native-CPU kernels do), so the plugin's Where passes
 * the condition tensor's dtype and a Less -> Where chain behaves the same on either convention. */
int infini_rocm_where_ex(infiniRocmRuntime_t rt, int dtype, int cond_dtype, const void *x, const void *y,
                         const void *cond, void *out, int ndim, const int64_t *shape, const int64_t *stride_x,
                         const int64_t *stride_y, const int64_t *stride_c);
/* Pad (constant 0) and Slice in one kernel, like the reference (PadSliceCudaCompute,
 * src/kernels/cuda/pad_slice.cc:4-45): out[i] = in[starts + i*steps] when inside `in`, else 0.
 * Slice: starts >= 0 (steps honoured — the reference CUDA kernel ignores them, pad_slice.cc:35-40);
 * Pad: starts = -pads_begin, steps = 1 (NULL = all ones). */
int infini_rocm_pad_slice(infiniRocmRuntime_t rt, int dtype, const void *x, void *y, int ndim,
                          const int64_t *in_shape, const int64_t *out_shape, const int64_t *starts,
                          const int64_t *steps, int reserved);
/* 2-D strided byte copy: `rows` rows of `row_bytes`, pitches in bytes. Concat / Split are one call per
 * input / output (reference: ConcatCuda / SplitCuda, src/kernels/cuda/split_concat.cu:29-82). */
int infini_rocm_strided_copy(infiniRocmRuntime_t rt, const void *src, void *dst, int64_t rows,
                             int64_t row_bytes, int64_t src_pitch, int64_t dst_pitch);
/* `count` 2-D strided copies of the same `rows` as ONE launch (blockIdx.z = segment; more than 16 segments: one launch per 16): a Concat
 * is one segment per input, a Split one per output. Segments with row_bytes == 0 are skipped. */
int infini_rocm_strided_copy_multi(infiniRocmRuntime_t rt, int count, const void *const *srcs, void *const *dsts, int64_t rows,
                                   const int64_t *row_bytes, const int64_t *src_pitch, const int64_t *dst_pitch);

/* ------------------------------------------------------------------------------------------ */
/* "From-shape" forms (csrc/shaped.hip): the operators above whose arguments are strides / extents, taking the operands' SHAPES as the */
/*   reference's operators hold them (TensorObj::getDims()). The shape -> stride glue lives once, below the ABI: the plugin kernels      */
/*   (plugin/src/rocm_kernels.cc) and the ctypes mirror (infinitensor_amd/ops.py) both call these and nothing else for these operators. */
/*   Reference glue they replace: the kernels' own broadcast index arithmetic (src/kernels/cuda/element_wise.cu:9-60, where.cu),        */
/*   matmul.cc:86-137 (batch broadcast by zero stride, bias expanded to [.., m, n]), split_concat.cc:10-62, pad_slice.cc, gather.cc.    */
/* ------------------------------------------------------------------------------------------ */
/* strides[out_rank]: element strides of a dense tensor of `shape` viewed in `out_shape`, 0 where a dim is broadcast (a dim of 1 or a
 * missing leading dim); INVALID_ARGUMENT when the shape does not broadcast (infer_broadcast, src/utils/operator_utils.cc:6-32). Pure. */
int infini_rocm_broadcast_strides(int rank, const int64_t *shape, int out_rank, const int64_t *out_shape, int64_t *strides);
int infini_rocm_binary_shaped(infiniRocmRuntime_t rt, int op, int dtype, const void *a, int a_rank, const int64_t *a_shape,
                              const void *b, int b_rank, const int64_t *b_shape, void *out, int out_rank, const int64_t *out_shape);
int infini_rocm_where_shaped(infiniRocmRuntime_t rt, int dtype, int cond_dtype, const void *x, int x_rank, const int64_t *x_shape,
                             const void *y, int y_rank, const int64_t *y_shape, const void *cond, int c_rank, const int64_t *c_shape,
                             void *out, int out_rank, const int64_t *out_shape);
int infini_rocm_expand_shaped(infiniRocmRuntime_t rt, int dtype, const void *x, int x_rank, const int64_t *x_shape, void *out,
                              int out_rank, const int64_t *out_shape);
/* plan[9] = { batch, m, n, k, stride_a, stride_b, bias_batch_stride, bias_m_stride, bias_n_stride } of op(A) op(B) (+ bias broadcast to
 * [batch dims .., m, n]); bias_rank < 0: no bias. Errors as the reference's asserts: K mismatch, batch dims that do not broadcast, a
 * partially broadcast batch (matmul.cc:124-137). Pure. */
int infini_rocm_matmul_plan(int a_rank, const int64_t *a_shape, int b_rank, const int64_t *b_shape, int bias_rank,
                            const int64_t *bias_shape, int trans_a, int trans_b, int64_t *plan);
/* infini_rocm_matmul_headsplit on that plan (bias == NULL: none; seq = head_dim = 0: the plain [.., m, n] store). */
int infini_rocm_matmul_shaped(infiniRocmRuntime_t rt, int dtype, const void *a, int a_rank, const int64_t *a_shape, const void *b,
                              int b_rank, const int64_t *b_shape, const void *bias, int bias_rank, const int64_t *bias_shape, void *out,
                              int trans_a, int trans_b, int act, int64_t seq, int64_t head_dim);
/* Concat: input i holds axis_extents[i] slices of `out`'s axis (extents add up to out_shape[axis]; empty inputs allowed) — one launch.
 * Split: output i receives axis_extents[i] slices of `in`'s axis. elem_size in bytes. */
int infini_rocm_concat_shaped(infiniRocmRuntime_t rt, int elem_size, int count, const void *const *inputs, const int64_t *axis_extents,
                              void *out, int out_rank, const int64_t *out_shape, int axis);
int infini_rocm_split_shaped(infiniRocmRuntime_t rt, int elem_size, int count, void *const *outputs, const int64_t *axis_extents,
                             const void *in, int in_rank, const int64_t *in_shape, int axis);
/* Constant-0 Pad; pads = begin_0 .. begin_{rank-1}, end_0 .. end_{rank-1} (include/operators/pad.h), y has in_shape[d] + begin_d + end_d. */
int infini_rocm_pad_shaped(infiniRocmRuntime_t rt, int dtype, const void *x, void *y, int rank, const int64_t *in_shape,
                           const int64_t *pads);
/* Gather along `axis` of data_shape with n_indices indices (output rank = data rank - 1 + index rank, include/operators/gather.h:27-49). */
int infini_rocm_gather_shaped(infiniRocmRuntime_t rt, int dtype, int index_dtype, const void *data, int data_rank,
                              const int64_t *data_shape, const void *indices, int64_t n_indices, void *out, int axis);

/* ------------------------------------------------------------------------------------------ */
/* RCCL communicator + collectives (reference: NcclCommunicatorObj, include/cuda/nccl_communicator.h:22-68; */
/*   CudaRuntimeObj::initComm, src/cuda/cuda_runtime.cc:495-509; kernels all_reduce.cc:8-63,        */
/*   all_gather.cc:8-40, broadcast.cc:8-26, send.cc:8-37, recv.cc:8-41). One communicator per runtime, */
/*   one process per GPU; every collective is enqueued on the runtime stream.                        */
/* ------------------------------------------------------------------------------------------ */
/* File-based rendezvous exactly like the reference: rank 0 writes ./<name>_nccl_id.bin, others poll
 * (100 ms, 10 s limit). */
int infini_rocm_comm_init(infiniRocmRuntime_t rt, const char *name, int world_size, int rank);
/* Rendezvous through a channel the launcher already has: rank 0 creates the id, every rank passes it. */
int infini_rocm_comm_unique_id(void *buf, size_t *nbytes);
int infini_rocm_comm_init_id(infiniRocmRuntime_t rt, const void *unique_id, size_t nbytes,
                             int world_size, int rank);
/* The hand-written one-hop transport (csrc/comm_direct.hip): collectives as HIP kernels pushing into IPC-mapped, uncached
 * peer buffers with flag signalling — the fully connected xGMI mesh of one MI355X node needs no ring and no RCCL
 * (reference counterpart: the same NcclCommunicatorObj / *NCCL kernels as above). Ranks may share a device, which RCCL
 * refuses: the reference's multi-rank tests then run on a one-GPU box. File rendezvous ./<name>_xgmi_<rank>.bin in the cwd;
 * every process needs HSA_ENABLE_IPC_MODE_LEGACY=0. world_size <= 8. infini_rocm_comm_init does the same when the
 * environment holds INFINI_ROCM_COMM=direct (how the plugin's init_comm selects it). Settings (equal on all ranks):
 * INFINI_ROCM_DIRECT_CAP_MB (8: bytes per box slot; a message of more than world x cap goes in pieces),
 * INFINI_ROCM_DIRECT_TIMEOUT_S (600: a kernel gives up waiting for a peer, sets the error word and terminates — it never
 * hangs the GPU; the next infini_rocm_runtime_sync (or infini_rocm_comm_check) returns INFINI_ROCM_RCCL_ERROR ONCE and clears the
 * word: the tensors of the collectives since the previous sync are undefined, later collectives wait again).
 * One process per rank (an IPC handle cannot be opened by its exporter). A synchronous collective issued while *_async ones are
 * pending first joins the comm stream (the two families share the communicator's sequence numbers). */
int infini_rocm_comm_init_direct(infiniRocmRuntime_t rt, const char *name, int world_size, int rank);
/* Which transport the collectives use when both are initialised: 0 RCCL (default), 1 the direct transport. */
int infini_rocm_comm_set_algo(infiniRocmRuntime_t rt, int algo);
/* Blocking (one 4-byte device read): INFINI_ROCM_RCCL_ERROR when a direct-transport kernel ran into its time limit; clears the error. */
int infini_rocm_comm_check(infiniRocmRuntime_t rt);
int infini_rocm_comm_destroy(infiniRocmRuntime_t rt);
int infini_rocm_comm_info(infiniRocmRuntime_t rt, int *world_size, int *rank);
/* op: 0 sum, 1 prod, 2 min, 3 max, 4 avg; count in elements; in-place allowed (x == y). */
int infini_rocm_all_reduce(infiniRocmRuntime_t rt, int op, int dtype, const void *x, void *y,
                           int64_t count);
/* y receives world_size * count elements, rank r's block at offset r * count. */
/* Overlapped collectives. infini_rocm_all_reduce_async is ordered after everything enqueued on the runtime stream so far
 * but runs on a second, runtime-owned stream: later work on the runtime stream does NOT wait for it until
 * infini_rocm_comm_join. A row-parallel GEMM cut into row chunks can so hide each chunk's all-reduce under the next chunk's
 * GEMM (the reference issues ONE whole-tensor ncclAllReduce behind the whole GEMM: all_reduce.cc:10-33). Capturable (the
 * fork / join become edges of the hipGraph); every rank must issue its collectives in the same order. */
int infini_rocm_all_reduce_async(infiniRocmRuntime_t rt, int op, int dtype, const void *x, void *y, int64_t count);
int infini_rocm_comm_join(infiniRocmRuntime_t rt);
/* y[count] = sum over ranks r of x_r[rank * count ..]: the reduce-scatter half of an all-reduce (x: world_size * count
 * elements). direct != 0: one-hop exchange over the fully connected xGMI mesh (grouped send / recv of every slice to its
 * owner, then a local fp32 sum) instead of RCCL's own algorithm choice; scratch from the runtime workspace. With
 * infini_rocm_all_gather this is the sequence-parallel form of a TP block: reduce-scatter -> residual add + norm on the
 * shard -> all-gather. No reference counterpart (the reference has AllReduce / AllGather only). */
int infini_rocm_reduce_scatter(infiniRocmRuntime_t rt, int dtype, const void *x, void *y, int64_t count, int direct);
int infini_rocm_all_gather(infiniRocmRuntime_t rt, int dtype, const void *x, void *y, int64_t count);
int infini_rocm_broadcast(infiniRocmRuntime_t rt, int dtype, const void *x, void *y, int64_t count,
                          int root);
int infini_rocm_send(infiniRocmRuntime_t rt, int dtype, const void *x, int64_t count, int peer);
int infini_rocm_recv(infiniRocmRuntime_t rt, int dtype, void *y, int64_t count, int peer);

#ifdef __cplusplus
}
#endif
#endif /* INFINI_ROCM_H */
