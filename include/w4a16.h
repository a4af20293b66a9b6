/*
 * w4a16.h - C ABI of libw4a16.so: AWQ 4-bit-weight x fp16-activation GEMM for
 * AMD MI355X (gfx950).
 *
 * Plain pointers and sizes only; every pointer is a DEVICE pointer unless a
 * comment says otherwise.  No entry point allocates, frees or synchronises:
 * each one enqueues its kernels on `stream` (a hipStream_t, NULL = the default
 * stream) and returns, so all of them may be called under HIP graph capture.
 * The current HIP device must be the one that owns the buffers.
 *
 * Tensor conventions (identical to the reference's AWQ tensors):
 *   x        [M, K]       fp16, row-major, contiguous
 *   qweight  [K, N/8]     int32; column 8c+j of row k is nibble {0,4,1,5,2,6,3,7}[j] of word c
 *   qzeros   [K/G, N/8]   int32, same nibble order
 *   scales   [K/G, N]     fp16
 *   out      [M, N]       fp16
 * w[k][n] = fp16((q[k][n] - z[k/G][n]) * s[k/G][n]), one rounding, exactly as
 * the reference dequantises; products accumulate in fp32 and are rounded to
 * fp16 once.
 *
 * Return value: 0 on success, a negative W4A16_E* code otherwise.  Nothing is
 * enqueued when a code is returned for an argument error.
 */
#ifndef W4A16_H_
#define W4A16_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define W4A16_ABI_VERSION 1

enum {
  W4A16_OK = 0,
  W4A16_E_OC_PACK = -1,   /* "OC is not multiple of pack_num = 8"   (gemm_kernels.cu:414-415) */
  W4A16_E_GROUP32 = -2,   /* "Group size should be a multiple of 32" (gemm_kernels.cu:416-417) */
  W4A16_E_K_GROUP = -3,   /* K is not a multiple of the group size   (awq.py:84-88)            */
  W4A16_E_NULL = -4,      /* a required pointer is NULL                                         */
  W4A16_E_SHAPE = -5,     /* M, K, N or G is not positive, or K is not a multiple of 32         */
  W4A16_E_ALIGN = -6,     /* a pointer is not 16-byte aligned                                   */
  W4A16_E_LAUNCH = -7,    /* the HIP runtime refused the launch; see w4a16_last_hip_error()     */
  W4A16_E_TOO_LARGE = -8, /* an index would overflow 32 bits                                    */
  W4A16_E_WORKSPACE = -9  /* the workspace is smaller than the launch shape asked for needs     */
};

/* Human-readable text for a return code (host pointer, static storage). */
const char* w4a16_strerror(int code);

/* hipError_t of the last failed launch on this host thread (0 if none). */
int w4a16_last_hip_error(void);

/* Returns W4A16_ABI_VERSION of the loaded library. */
int w4a16_abi_version(void);

/*
 * out = x @ dequant(qweight, qzeros, scales), on the stock AWQ tensors.
 * Replaces:  torch::Tensor awq_gemm(_in_feats, _kernel, _scaling_factors, _zeros, split_k_iters)
 *            csrc/quantization/awq/gemm_kernels.cu:389-446 (declared csrc/ops.h:69-74,
 *            bound csrc/pybind.cpp:52-55).
 * `split_k_iters` is accepted for signature parity and ignored (the reference
 * uses it only to size its [split_k, M, N] scratch tensor and the trailing sum).
 * Unlike the reference this accepts any N % 8 == 0 (the reference needs N % 64 == 0,
 * which excludes 1376-wide tensor-parallel shards).
 * This entry point reads the stock layout directly and is the shape-agnostic
 * path; the tuned path is w4a16_repack() once + w4a16_gemm_packed_f16().
 */
int w4a16_gemm_f16(const void* x, const void* qweight, const void* scales, const void* qzeros,
                   void* out, int M, int K, int N, int G, int split_k_iters, void* stream);

/*
 * out[K, N] fp16 = dequant(qweight, qzeros, scales).
 * Replaces:  torch::Tensor awq_dequantize(_kernel, _scaling_factors, _zeros, split_k_iters, thx, thy)
 *            csrc/quantization/awq/gemm_kernels.cu:332-381 (declared csrc/ops.h:76-82).
 * The reference's split_k_iters/thx/thy are launch hints with no effect on the
 * result and have no counterpart here.
 */
int w4a16_dequantize_f16(const void* qweight, const void* scales, const void* qzeros, void* out,
                         int K, int N, int G, void* stream);

/*
 * Sizes in bytes of the private MFMA-ordered buffers w4a16_repack() fills.
 * Pure host arithmetic.  Both return 0 for invalid shapes.
 * No counterpart in the reference: its kernel consumes the stock layout and
 * pays for that with a shared-memory round trip of the dequantised weights
 * (gemm_kernels.cu:139-169).
 */
size_t w4a16_packed_weight_bytes(int K, int N);
size_t w4a16_packed_sz_bytes(int K, int N, int G);

/*
 * One-time re-layout of the stock tensors into the private format
 *   packed_w  : [ceil(N/16)][KB][64 lanes][4] uint32, KB = ceil(K/512)*4 - word j of lane l of block
 *               (nt, kb) holds k = 128 kb + 32 j + 8 (l >> 4) + i (i = 0..7, nibble order
 *               {0,4,1,5,2,6,3,7}) of column n = 16 nt + (l & 15): one 16-byte load per lane
 *               is four ready-made 16x16x32 MFMA A-operand fragments.
 *   packed_sz : [ceil(N/16)][K/G][16] uint32 - low half fp16 scale, high half fp16 -(1024 + zero).
 * Padding columns (n >= N) are zero words with zero scale; padding k (k >= K) repeats the last
 * group's zero point, i.e. dequantises to exactly 0.  The stock tensors are not modified.
 */
int w4a16_repack(const void* qweight, const void* scales, const void* qzeros, void* packed_w,
                 void* packed_sz, int K, int N, int G, void* stream);

/*
 * out = x @ W (+ bias) from the private format.  `bias` ([N] fp16) may be NULL.
 * This is the kernel behind awq_gemm / AWQLinearMethod.apply_weights
 * (vllm/model_executor/layers/quantization/awq.py:143-165) once a layer's
 * weights have been repacked; the optional bias folds awq.py:163-164 into the
 * epilogue.
 */
int w4a16_gemm_packed_f16(const void* x, const void* packed_w, const void* packed_sz,
                          const void* bias, void* out, int M, int K, int N, int G, void* stream);

/*
 * Same as w4a16_gemm_packed_f16 with an explicit launch shape of the streaming kernel, for tuning
 * and tests: mt = 16-row output tiles per workgroup (1, 2, 4 or 8), t = 16-column tiles per
 * workgroup (1..8, mt * t <= 24), waves = wavefronts per workgroup (4, 8 or 16).  Only the
 * combinations the library was built with are accepted (W4A16_FOR_EACH_CFG in w4a16.hip); others
 * return W4A16_E_SHAPE.
 */
int w4a16_gemm_packed_f16_cfg(const void* x, const void* packed_w, const void* packed_sz,
                              const void* bias, void* out, int M, int K, int N, int G, int mt,
                              int t, int waves, void* stream);

/*
 * The LDS-staged variant of the same GEMM with an explicit launch shape, for tuning and tests
 * (w4a16_gemm_packed_f16 picks it by itself when it is the faster one): a workgroup owns
 * mt x (nw * t) 16x16 output tiles; its nw * kw wavefronts form kw groups that split K, the nw
 * wavefronts of a group split the columns and share the group's activation slab through LDS.
 * Built combinations only; others return W4A16_E_SHAPE.
 */
int w4a16_gemm_packed_f16_tiled_cfg(const void* x, const void* packed_w, const void* packed_sz,
                                    const void* bias, void* out, int M, int K, int N, int G, int mt,
                                    int t, int nw, int kw, void* stream);

/*
 * The variant built on 32x32x16 matrix instructions, with an explicit launch shape, for tuning and
 * tests (w4a16_gemm_packed_f16 picks it by itself when it is the faster one): a workgroup owns
 * mt x t tiles of 32 rows x 32 columns; its kw wavefronts split K and each stages the activations of
 * its own K blocks in a private piece of LDS (no barrier in the main loop).  Built combinations only;
 * others return W4A16_E_SHAPE.  kw < 0 selects the variant for many rows instead: a workgroup of nw = -kw
 * wavefronts owns mt x (t * nw) such tiles over the whole of K, each wavefront t column tiles, all sharing
 * one activation slab per 128-k block (a ring of three in LDS, one barrier per block, no reduction).
 * In the `_cfg` / `_ex` entry points below these kernels are selected by (mt, t, a = 1, b = -kw) and
 * (mt, t, a = nw, b = -1).
 */
int w4a16_gemm_packed_f16_wide_cfg(const void* x, const void* packed_w, const void* packed_sz,
                                   const void* bias, void* out, int M, int K, int N, int G, int mt,
                                   int t, int kw, void* stream);

/*
 * `batch` independent GEMMs of one shape in a single launch: problem b computes
 *   out + b * out_batch_stride  =  (x + b * x_batch_stride) @ W_b  (+ bias + b * N)
 * with W_b at packed_w + b * w4a16_packed_weight_bytes(K, N) and packed_sz +
 * b * w4a16_packed_sz_bytes(K, N, G): the buffers of the problems stand back to back.
 * Strides are in fp16 elements; x_batch_stride == 0 gives every problem the same activations;
 * x_batch_stride must be a multiple of 8, out_batch_stride a multiple of 4 and >= M * N.
 * Replaces the per-expert loop of the reference's quantised Mixtral block, which runs every
 * local expert's w1 / w3 / w2 AWQ GEMM on all tokens one launch at a time
 * (vllm/model_executor/models/mixtral_quant.py:87-93, 145-157): 16 gate/up problems share the
 * tokens (stride 0), the 8 down projections take one activation matrix each.
 */
int w4a16_gemm_packed_f16_batched(const void* x, long long x_batch_stride, const void* packed_w,
                                  const void* packed_sz, const void* bias, void* out,
                                  long long out_batch_stride, int M, int K, int N, int G, int batch,
                                  void* stream);

/*
 * w4a16_gemm_packed_f16_batched with an explicit launch shape, for tuning and tests:
 * (mt, t, a, 0) = the streaming kernel with a waves (see w4a16_gemm_packed_f16_cfg),
 * (mt, t, a, b) with b > 0 = the LDS-staged kernel with nw = a, kw = b,
 * (mt, t, 1, -kw) = the 32x32x16 kernel with private slabs, (mt, t, nw, -1) with nw > 1 = the one with a shared slab.
 */
int w4a16_gemm_packed_f16_batched_cfg(const void* x, long long x_batch_stride, const void* packed_w,
                                      const void* packed_sz, const void* bias, void* out,
                                      long long out_batch_stride, int M, int K, int N, int G,
                                      int batch, int mt, int t, int a, int b, void* stream);

/*
 * The general form: batched (batch >= 1, strides as for w4a16_gemm_packed_f16_batched; ignored for
 * batch == 1) with a fused epilogue and an optional explicit launch shape (mt == 0: chosen by the
 * library; otherwise as for w4a16_gemm_packed_f16_batched_cfg).
 *   W4A16_EPI_NONE      out [M, N]   = x @ W (+ bias)
 *   W4A16_EPI_SILU_MUL  out [M, N/2] = silu(g) * u, where [g | u] = x @ W (+ bias) are the two halves
 *                       of the N columns (N a multiple of 32): the merged gate_up projection of a
 *                       gated MLP followed by its activation.  Replaces the GEMM + `silu_and_mul`
 *                       pair of vllm/model_executor/models/llama.py:74-80
 *                       (csrc/activation_kernels.cu:9-34): silu in fp32 on the fp16 GEMM result,
 *                       rounded to fp16, then an fp16 multiply, as there; the [M, N] intermediate
 *                       never reaches memory.  Built launch shapes only (W4A16_E_SHAPE otherwise).
 *   W4A16_EPI_ADD_TO_OUT  out [M, N] = out + (x @ W (+ bias)): the fp16 GEMM result (rounded once, bias
 *                       added in fp16) is added in fp16 to what `out` holds, i.e. the
 *                       `residual + linear(x)` of a decoder layer (vllm/model_executor/models/llama.py:
 *                       the o_proj and down_proj outputs are added to the residual stream) with `out` the
 *                       residual buffer, updated in place; bit-identical to the two separate steps.
 * out_batch_stride >= M * (columns of out).
 */
enum { W4A16_EPI_NONE = 0, W4A16_EPI_SILU_MUL = 1, W4A16_EPI_ADD_TO_OUT = 2 };
int w4a16_gemm_packed_f16_ex(const void* x, long long x_batch_stride, const void* packed_w,
                             const void* packed_sz, const void* bias, void* out,
                             long long out_batch_stride, int M, int K, int N, int G, int batch,
                             int epilogue, int mt, int t, int a, int b, void* stream);

/*
 * bf16 activations on the same packed weights: out [M, N] (bf16) = x [M, K] (bf16) @ W (+ bias, bf16), fp32
 * accumulation, with W = bf16((q - z) * s) rounded once from the exact product (scales as the checkpoint holds
 * them, fp16).  No counterpart in the reference, whose AWQ path accepts fp16 only
 * (vllm/model_executor/layers/quantization/awq.py:42-43): there is no reference result to compare with, the tests
 * check against a float64 restatement of the definition above.  Epilogues: W4A16_EPI_NONE, W4A16_EPI_ADD_TO_OUT
 * (bf16 + bf16, rounded once); others return W4A16_E_SHAPE.  mt == 0: the library's launch shape (up to 16 rows
 * the decode kernel where K is a multiple of 128 (up to 32 rows on narrow layers, as for fp16), else the streaming
 * 16x16x32 kernel; beyond that the 32x32x16
 * kernels where K is a multiple of 128: private slabs, and beyond 256 rows the shared slab where fp16 takes it);
 * mt > 0: (mt, t, waves) of the streaming kernel as built for bf16; mt == -16: (t, waves) of the decode kernel (up
 * to 16 rows, K a multiple of 128); mt == -36: (t, nw = waves) of the shared-slab 32x32x16 kernel with four row
 * tiles (K a multiple of 128, at least 256); other mt < 0: (-mt, t, kw = waves) of the private-slab 32x32x16 kernel as
 * built for bf16 (all listed by w4a16_built_shape_bf16); W4A16_E_SHAPE otherwise.
 */
int w4a16_gemm_packed_bf16(const void* x, const void* packed_w, const void* packed_sz, const void* bias, void* out,
                           int M, int K, int N, int G, int epilogue, int mt, int t, int waves, void* stream);
/* The i-th built (mt, t, waves) of the bf16 kernels in shape[3] (mt == -16: the decode kernel, -36: the shared-slab
 * 32x32x16 kernel, other mt < 0: the private-slab one); W4A16_E_SHAPE past the end. */
int w4a16_built_shape_bf16(int i, int* shape);

/*
 * w4a16_gemm_packed_f16_ex with the library's launch shape and a per-problem switch: `problem_mask`
 * is a DEVICE array of `batch` int32 (NULL = all on); problems whose entry is 0 are skipped and their
 * part of `out` is left as it was.  The mask is read by the kernel, not by the host, so it can be the
 * output of an earlier kernel on the same stream (no host synchronisation, HIP-graph capturable).
 * `active_hint`: how many problems the caller expects to be on (0 = unknown); it only steers the
 * launch shape, which is otherwise sized for `batch` busy problems.
 * For a mixture-of-experts block at a few tokens: only the experts some token was routed to are
 * read from memory, where the reference runs every local expert on every token and multiplies the
 * unselected ones by zero (vllm/model_executor/models/mixtral_quant.py:145-157).
 */
int w4a16_gemm_packed_f16_masked(const void* x, long long x_batch_stride, const void* packed_w,
                                 const void* packed_sz, const void* bias, void* out,
                                 long long out_batch_stride, int M, int K, int N, int G, int batch,
                                 int epilogue, const void* problem_mask, int active_hint, void* stream);

/*
 * The decode kernel (up to 16 rows, and 17-32 rows of the plain GEMM with t = 1 and 16 waves; the K split up to 4
 * rows and 4 tiles or tile pairs per workgroup) with an
 * explicit launch shape, for tuning and tests;
 * batching and epilogues as for w4a16_gemm_packed_f16_ex.  A workgroup owns t 16-column tiles (fused activation: t / 2 gate + t / 2 up
 * tiles) for the rows; its `waves` wavefronts split the workgroup's share of K by 128-k blocks, each keeps
 * `depth` blocks of weights in flight (1 or 2; 0 = the library's choice) and, up to 4 rows, stages its activations
 * in a private piece of LDS (5-32 rows: a ring of LDS slots filled block by block, or fragments straight from L2 where
 * the ring does not fit).  splits > 1 divides K over that many workgroups per output slot: each leaves fp32 partial sums in
 * `workspace`, and the one that arrives last adds them in split order, so the result does not depend on timing.
 * `workspace` (16-byte aligned; may be NULL when splits == 1) must hold w4a16_decode_workspace_bytes(...) bytes,
 * zero before its first use: the kernels return every counter they touch to zero, so one buffer serves any
 * sequence of launches, but only one launch at a time.  Built (t, waves) only, W4A16_E_SHAPE otherwise;
 * W4A16_E_WORKSPACE if the buffer is too small.  No counterpart in the reference, whose split over K goes
 * through a [split_k, M, N] tensor and a separate sum (gemm_kernels.cu:420-445).
 */
int w4a16_gemm_packed_f16_decode_cfg(const void* x, long long x_batch_stride, const void* packed_w,
                                     const void* packed_sz, const void* bias, void* out,
                                     long long out_batch_stride, int M, int K, int N, int G, int batch,
                                     int epilogue, int t, int waves, int depth, int splits, void* workspace,
                                     size_t workspace_bytes, void* stream);
size_t w4a16_decode_workspace_bytes(int K, int N, int batch, int epilogue, int t, int waves, int splits);

/*
 * The 32x32x16 kernel with private slabs (w4a16_gemm_packed_f16_wide_cfg, kw > 0) with K split over `splits`
 * workgroups per output tile, for tuning and tests (w4a16_gemm_packed_f16_ws picks it by itself for 97-128 rows):
 * every workgroup of a tile leaves its fp32 partial sums in its own slot of `workspace`, draws a ticket, and the
 * one that arrives last adds the slots in split order and stores, so the result does not depend on timing.
 * With two workgroups per tile a 128-row x 96-column tile (mt = 4, t = 3, kw = 4) fills the chip on the
 * 4096 x 11008 layer at 128 rows, and one converted weight fragment feeds four matrix instructions instead of two.
 * splits > 1 needs a 4-wave shape and a `workspace` (16-byte aligned) of w4a16_wide_workspace_bytes(...) bytes,
 * zero before its first use; same rules for the buffer as for w4a16_gemm_packed_f16_decode_cfg, whose buffer it
 * may share.  Epilogues as for w4a16_gemm_packed_f16_ex (one problem).  No counterpart in the reference, whose
 * split over K goes through a [split_k, M, N] fp16 tensor and a separate sum (gemm_kernels.cu:420-445).
 */
int w4a16_gemm_packed_f16_wide_split_cfg(const void* x, const void* packed_w, const void* packed_sz, const void* bias,
                                         void* out, int M, int K, int N, int G, int epilogue, int mt, int t, int kw,
                                         int splits, void* workspace, size_t workspace_bytes, void* stream);
size_t w4a16_wide_workspace_bytes(int M, int N, int batch, int epilogue, int mt, int t, int splits);

/*
 * w4a16_gemm_packed_f16_ex with the library's launch shape and an optional workspace (NULL: exactly
 * w4a16_gemm_packed_f16_ex with mt == 0).  With a workspace of at least w4a16_workspace_bytes(...) bytes the
 * library may split K over workgroups where the plain launch would leave most of the chip idle (few columns,
 * long K, up to 4 rows: the per-rank qkv projection of a tensor-parallel model at decode) or short of matrix
 * work per converted weight (97-128 rows: 128-row tiles, two workgroups per tile).  Same rules for the
 * buffer as above; a buffer that is too small is ignored.  w4a16_workspace_bytes returns 0 where no split is
 * chosen.  w4a16_launch_shape_ws reports the choice made with a workspace (W4A16_KERNEL_DECODE: (1, t, waves,
 * splits); W4A16_KERNEL_MFMA32: (mt, t, splits, kw)).
 */
size_t w4a16_workspace_bytes(int M, int K, int N, int G, int batch, int epilogue);
int w4a16_gemm_packed_f16_ws(const void* x, long long x_batch_stride, const void* packed_w, const void* packed_sz,
                             const void* bias, void* out, long long out_batch_stride, int M, int K, int N, int G,
                             int batch, int epilogue, void* workspace, size_t workspace_bytes, void* stream);
int w4a16_launch_shape_ws(int M, int K, int N, int G, int batch, int epilogue, int shape[5]);

/*
 * w4a16_gemm_packed_f16_decode_cfg without the workspace arguments: the K-split launch runs on a device array of
 * w4a16_static_scratch_bytes() bytes (1 MiB) that is part of the library's code object, so this entry point calls
 * no allocator at all - neither hipMalloc / hipFree nor any routine that allocates.  No counterpart in the
 * reference; it is for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses
 * that forbid allocation.  t == 0 takes the launch shape the library picks when it has a workspace (waves, depth
 * and splits are then ignored; W4A16_E_SHAPE if the problem is not one for the decode kernel, i.e. more than 16
 * rows or a shape it does not admit).  A launch that needs more than the array holds
 * (w4a16_decode_workspace_bytes(...) > w4a16_static_scratch_bytes()) is refused with W4A16_E_WORKSPACE and
 * nothing is enqueued.  Every GPU holds its own copy of the array until the process ends, and every call on a
 * GPU shares that copy: the kernels return the counters they used to zero, so calls on one GPU may follow each
 * other on any streams, but two that OVERLAP on one GPU race.  The current device must be the one that owns the
 * buffers (the array's address is looked up, once per device, for the current device).
 */
size_t w4a16_static_scratch_bytes(void);
int w4a16_gemm_packed_f16_decode_static(const void* x, long long x_batch_stride, const void* packed_w,
                                        const void* packed_sz, const void* bias, void* out,
                                        long long out_batch_stride, int M, int K, int N, int G, int batch,
                                        int epilogue, int t, int waves, int depth, int splits, void* stream);

/*
 * Which kernel and launch shape w4a16_gemm_packed_f16 / _batched / _ex pick by themselves for a problem
 * (pure host arithmetic; for benchmarks, profiles and tests that need to name what ran).  No counterpart
 * in the reference, whose launch shape is fixed (gemm_kernels.cu:422-444).
 *   shape[0]  kernel: one of W4A16_KERNEL_*
 *   shape[1..4]  STREAMING   (mt, t, waves, 0)      16-row / 16-column tiles, `waves` wavefronts split K
 *                LDS_STAGED  (mt, t, nw, kw)        16-row / 16-column tiles
 *                MFMA32      (mt, t, splits, kw)    32-row / 32-column tiles, private slabs per wavefront
 *                                                   (splits = 1 unless a workspace lets K be split over workgroups)
 *                MFMA32_SHARED (mt, t, nw, 1)       32-row / 32-column tiles, one slab per workgroup
 *                DECODE      (1, t, waves, splits)  up to 16 rows (32 with t = 1 and 16 waves)
 * w4a16_launch_shape takes no group size: it answers for a group size the decode kernel admits (32, 64 or a
 * multiple of 128, i.e. every AWQ checkpoint in use); for another multiple of 32 the launches it reports as
 * DECODE are made by the streaming kernel with the same (t, waves).  w4a16_launch_shape_ws takes the group size.
 */
enum { W4A16_KERNEL_STREAMING = 0, W4A16_KERNEL_LDS_STAGED = 1, W4A16_KERNEL_MFMA32 = 2, W4A16_KERNEL_MFMA32_SHARED = 3,
       W4A16_KERNEL_DECODE = 4 };
int w4a16_launch_shape(int M, int K, int N, int batch, int epilogue, int shape[5]);

/*
 * Enumerates the launch shapes this build holds of kernel `kernel` (W4A16_KERNEL_*): shape number `index`
 * (0, 1, ..) in the order of the arguments the _cfg entry points take - STREAMING (mt, t, waves),
 * LDS_STAGED (mt, t, nw, kw), MFMA32 (mt, t, kw), MFMA32_SHARED (mt, t, nw), DECODE (t, waves); W4A16_E_SHAPE past
 * the last one.
 * Lets a test suite run every shape that exists instead of a list kept by hand.  No counterpart in the reference.
 */
int w4a16_built_shape(int kernel, int index, int shape[4]);

/*
 * Measurement aid: one pass of 16-byte non-temporal loads over `bytes` bytes at `src` (16-byte aligned),
 * nothing else; `sink` is a device buffer of at least 8 KiB that is (practically) never written.  Its
 * duration on a buffer of a layer's packed size is the floor for that layer's decode launch on the same
 * machine.  No counterpart in the reference.
 */
int w4a16_stream_read(const void* src, size_t bytes, void* sink, void* stream);

/*
 * out[K, N] fp16 from the private format (bit-identical to w4a16_dequantize_f16
 * on the tensors the buffers were packed from).
 */
int w4a16_dequantize_packed_f16(const void* packed_w, const void* packed_sz, void* out, int K,
                                int N, int G, void* stream);

/*
 * Expert routing of a Mixtral-style block in one launch (vllm/model_executor/models/mixtral_quant.py:137-150:
 * softmax over all experts in fp32, top-k, renormalise by the sum of the selected, scatter): router_logits
 * [tokens, n_experts] fp16 -> weights [tokens, n_local] fp32 for this rank's experts
 * [first_local, first_local + n_local) (0 where a token did not select the expert; ties go to the lower index)
 * and, if `mask` is not NULL, mask[e] = 1 if any token selected local expert e, else 0 (int32 [n_local]: the
 * problem mask of w4a16_gemm_packed_f16_masked).  n_experts <= 64.
 */
int w4a16_moe_route_f16(const void* router_logits, void* weights, void* mask, int tokens, int n_experts,
                        int top_k, int first_local, int n_local, void* stream);

/*
 * The weighted sum over experts that ends the block (mixtral_quant.py:152-157: `current_hidden_states.mul_(
 * expert_weights)`, then the sum): out[t, h] = fp16(sum_e float(fp16(float(y[e, t, h]) * weights[t, e]))), e in
 * expert order, y [n_local, tokens, hidden] fp16 (the batched launch's output), weights as w4a16_moe_route_f16
 * writes them.  Experts whose weight for a token is 0 are not read: a masked launch's untouched outputs never
 * reach the sum.  hidden % 8 == 0; y and out 16-byte aligned.
 */
int w4a16_moe_combine_f16(const void* y, const void* weights, void* out, int n_local, int tokens, int hidden,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* W4A16_H_ */
