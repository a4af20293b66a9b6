// AWQ W4A16 GEMM for gfx950 (MI355X): kernels and the C ABI of libw4a16.so.
//
// Design
// ------
// The stock AWQ tensor packs eight adjacent OUTPUT columns of one k into a
// dword.  A gfx950 fp16 MFMA operand wants eight adjacent k of one column per
// lane.  w4a16_repack() therefore re-orders the weights once into 1-KiB blocks
// of 16 columns x 128 k in which lane l's 16 bytes are, after the usual
// magic-number int4->fp16 conversion, four ready-made A-operand fragments of
// v_mfma_f32_16x16x32_f16 (weights are the A operand: the 16 rows of the tile
// are output channels, so no matrix-core row is wasted on padding when M is
// small, and each lane ends up holding 4 adjacent output columns of one
// token, which store as 8 contiguous bytes).
//
// The GEMM kernels share that layout and the bit-exact register conversion
// (fp16((q - z) * s), as the reference computes it):
//
//  * gemm_decode_kernel (up to 16 rows): weights in 1-KiB units paced one or two
//    128-k blocks ahead per wave; up to 4 rows the activations are staged once per
//    wave in private LDS and K can be split over workgroups through a workspace
//    (see its comment).
//  * gemm_packed_kernel ("streaming", up to 32 rows): a workgroup owns T column
//    tiles x MT row tiles of the output over the WHOLE of K; its W wavefronts
//    split K between them (wave w takes 128-k blocks w, w+W, ...), stream their
//    weight blocks HBM -> VGPR with 16-byte loads, read the activation fragments
//    straight from L2, and combine their fp32 partial tiles through LDS at the end.
//  * gemm_tiled_kernel ("LDS-staged"): the same K split, but the activation slab
//    of each 128-k block is fetched once per workgroup in whole cache lines and
//    shared through swizzled LDS by the waves that split the workgroup's columns.
//  * gemm_wide_kernel / gemm_wide_shared_kernel (v_mfma_f32_32x32x16_f16, more
//    rows): activation slabs by LDS-DMA, private per wave with a K split, or one
//    per workgroup without.
//  * gemm_stock_kernel: the same arithmetic straight on the stock tensors.
//
// All of them take a batch of same-shaped problems in one launch (blockIdx.z);
// all but the last an optional fused silu(gate) * up epilogue for merged gate/up
// weights.  Only the decode kernel's K split goes across workgroups; its partial
// sums are added in split order, so no result depends on dispatch order.
//
// Reference behaviour replaced: csrc/quantization/awq/gemm_kernels.cu:30-272
// (GEMM), :274-381 (dequantize), csrc/quantization/awq/dequantize.cuh:17-84.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "../../include/w4a16.h"

// Compile-time ablation switches for kernel experiments (scripts/); 0 in every shipped build.
#ifndef W4A16_DBG
#define W4A16_DBG 0
#endif
#ifndef W4A16_PANEL
#define W4A16_PANEL 4
#endif
#ifndef W4A16_DEEP_TM
#define W4A16_DEEP_TM 1
#endif
#ifndef W4A16_HEAD
#define W4A16_HEAD 1
#endif
#ifndef W4A16_TMAJOR
#define W4A16_TMAJOR 1  // streaming kernel: tile-major arithmetic order within a block
#endif
#ifndef W4A16_SLIM
#define W4A16_SLIM 1    // streaming kernel: compact cross-wave reduction for up to 4 rows
#endif
#ifndef W4A16_WIDE_FROM
#define W4A16_WIDE_FROM 20  // the 32x32x16 kernel serves single problems with more rows than this
#endif
#ifndef W4A16_DECODE
#define W4A16_DECODE 1  // the decode kernel serves up to 4 rows by default
#endif
#ifndef W4A16_DECODE_STEAL
#define W4A16_DECODE_STEAL 1
#endif
#ifndef W4A16_ILV
#define W4A16_ILV 1  // the 32x32x16 kernel deals matrix and conversion instructions out alternately
#endif

namespace {

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Logical element i of a packed word sits in nibble kOrder[i].
__device__ __constant__ const int kOrderDev[8] = {0, 4, 1, 5, 2, 6, 3, 7};

__device__ __forceinline__ f16x2 as_h2(uint32_t v) { return __builtin_bit_cast(f16x2, v); }
__device__ __forceinline__ uint32_t as_u32(f16x2 v) { return __builtin_bit_cast(uint32_t, v); }

constexpr uint32_t kMagic = 0x64006400u;  // half2 {1024, 1024}
constexpr uint32_t kLoMask = 0x000f000fu;
constexpr uint32_t kHiMask = 0x00f000f0u;

// Per-(group, column) constants of the private format, expanded for packed math.
struct SZ {
  f16x2 s;    // {scale, scale}
  f16x2 nz1;  // {-(1024 + z)} x2 : turns (1024 + q) into q - z
  f16x2 nz2;  // {-(64 + z)} x2   : turns (1024 + 16 q) / 16 into q - z
};

__device__ __forceinline__ SZ expand_sz(uint32_t w) {
  const f16x2 v = as_h2(w);
  SZ r;
  r.s = f16x2{v.x, v.x};
  r.nz1 = f16x2{v.y, v.y};
  r.nz2 = r.nz1 + f16x2{(f16)960.0f, (f16)960.0f};
  return r;
}

// Constants of the int4 -> fp16 conversion, kept in registers the compiler cannot see through:
// gfx950 VOP3 takes no literal operand, so with compile-time constants `(w & mask) | magic`
// becomes v_and_b32 + v_or_b32 (two literals); from registers it is one v_and_or_b32.
struct DqConst {
  uint32_t lo, hi;  // nibble masks (SGPRs)
  uint32_t magic;   // half2 {1024, 1024} (VGPR: a VALU op reads at most one SGPR)
  f16x2 sixteenth;
};

__device__ __forceinline__ DqConst make_dq_const() {
  DqConst c;
  c.lo = kLoMask;
  c.hi = kHiMask;
  c.magic = kMagic;
  asm volatile("" : "+s"(c.lo), "+s"(c.hi));
  asm volatile("" : "+v"(c.magic));
  c.sixteenth = f16x2{(f16)0.0625f, (f16)0.0625f};
  return c;
}

// One packed word (8 k of one column) -> 8 fp16 weights in k order.
// q - z is formed exactly, then rounded once in the multiply by the scale:
// the same value as sub.f16x2 + fma.rn.f16x2 in gemm_kernels.cu:153-160.
// 13 VALU instructions per word: shift, 4 and-or, 2 add, 2 fma, 4 mul.
__device__ __forceinline__ f16x8 dequant_word(uint32_t w, const SZ& c, const DqConst& k) {
  const uint32_t t = w >> 8;
  const f16x2 h0 = as_h2((w & k.lo) | k.magic);
  const f16x2 h1 = as_h2((w & k.hi) | k.magic);
  const f16x2 h2 = as_h2((t & k.lo) | k.magic);
  const f16x2 h3 = as_h2((t & k.hi) | k.magic);
  const f16x2 d0 = (h0 + c.nz1) * c.s;
  const f16x2 d1 = (h1 * k.sixteenth + c.nz2) * c.s;
  const f16x2 d2 = (h2 + c.nz1) * c.s;
  const f16x2 d3 = (h3 * k.sixteenth + c.nz2) * c.s;
  return f16x8{d0.x, d0.y, d1.x, d1.y, d2.x, d2.y, d3.x, d3.y};
}

__device__ __forceinline__ f16x8 dequant_word(uint32_t w, const SZ& c) {
  const DqConst k = {kLoMask, kHiMask, kMagic, f16x2{(f16)0.0625f, (f16)0.0625f}};
  return dequant_word(w, c, k);
}

// out = silu(gate) * up on fp16 GEMM results, as the reference's activation kernel computes it
// (csrc/activation_kernels.cu:9-34): silu in fp32 on the fp16 value, rounded to fp16, then an
// fp16 multiply.
__device__ __forceinline__ f16x4 silu_mul(f16x4 g, f16x4 u) {
  f16x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x = (float)g[i];
    const f16 sg = (f16)(x / (1.0f + expf(-x)));
    r[i] = sg * u[i];
  }
  return r;
}

// ---------------------------------------------------------------------------
// GEMM on the private layout
// ---------------------------------------------------------------------------
// Waves whose partial tiles (1 KiB per 16x16 tile) take part in the final LDS round.
constexpr int kLdsTiles = 64;
constexpr int live_waves(int W, int TM) {
  int live = W;
  while (live > 1 && live * TM > kLdsTiles) live /= 2;
  return live;
}
// Tiles of LDS the reduction needs: the first fold holds W/2 sets, the final round LIVE sets.
constexpr int reduce_lds_tiles(int W, int TM) {
  if (W == 1) return 0;
  const int live = live_waves(W, TM);
  return (live < W ? W / 2 : live) * TM;
}

struct GemmArgs {
  const f16* x;
  const u32x4* pw;      // [NT][KB][64] x 16 bytes
  const uint32_t* psz;  // [NT][NG][16]
  const f16* bias;      // may be null
  f16* out;
  int M, K, N, G;
  int NT, KB, NG;
  int KBS;  // 128-k blocks per column tile in the packed layout (KB rounded up to a multiple of 4)
  int add_out;  // epilogue: out = out + fp16 result (the residual stream updated in place)
  const int* mask;  // batched launches: problems whose entry is 0 are skipped (out untouched); may be null
  // Batched launches (gridDim.z problems of one shape): element strides from one problem to the
  // next; all zero for a single GEMM.  x_bs == 0 shares the activations between the problems.
  long long x_bs, out_bs, pw_bs, psz_bs, bias_bs;
  // Worked out by the host once instead of by every wave (the waves of a workgroup share one scalar unit, and
  // an integer division is ~35 of its instructions):
  int batched;   // more than one problem in the launch (else the strides above are not even looked at)
  int g_magic;   // group sizes that are multiples of 128: group of block kb = (kb * g_magic) >> 16; 0: divide
  int mb_magic;  // kernels with row blocks, one panel: j / m_blocks = (j * mb_magic) >> 16; 0: divide
};

// Makes the compiler fetch the whole argument block in one batch of scalar loads at kernel entry.
// Left alone it fetches each field right before its first use, behind whatever branch precedes it:
// three to four dependent round trips to the kernel-argument buffer (a few hundred cycles each)
// before the first global load can be issued.
__device__ __forceinline__ void fetch_args_now(const GemmArgs& g) {
  asm volatile("" ::"s"(g.x), "s"(g.pw), "s"(g.psz), "s"(g.bias), "s"(g.out), "s"(g.M), "s"(g.K), "s"(g.N),
               "s"(g.G), "s"(g.NT), "s"(g.KB), "s"(g.NG), "s"(g.KBS), "s"(g.add_out), "s"(g.mask));
  asm volatile("" ::"s"(g.x_bs), "s"(g.out_bs), "s"(g.pw_bs), "s"(g.psz_bs), "s"(g.bias_bs), "s"(g.batched),
               "s"(g.g_magic), "s"(g.mb_magic));
}

// Group of k (a multiple of 32) for kernels whose scale/zero words cover SZS k-steps of 32.
template <int SZS>
__device__ __forceinline__ int group_of(const GemmArgs& a, int k) {
  if (SZS == 4 && a.g_magic != 0) return ((k >> 7) * a.g_magic) >> 16;
  return k / a.G;
}

// The argument block of problem blockIdx.z of a batched launch (scalar arithmetic only).
__device__ __forceinline__ GemmArgs batch_slice(const GemmArgs& g) {
  GemmArgs a = g;
  if (!g.batched) return a;
  const long long b = blockIdx.z;
  a.x = g.x + b * g.x_bs;
  a.pw = g.pw + b * g.pw_bs;
  a.psz = g.psz + b * g.psz_bs;
  // integer arithmetic: the host passes bias_bs = 0 with a null bias, which then stays null without a branch
  a.bias = reinterpret_cast<const f16*>(reinterpret_cast<uintptr_t>(g.bias) + (uintptr_t)(b * g.bias_bs * 2));
  a.out = g.out + b * g.out_bs;
  return a;
}

#ifdef W4A16_STAMPS
// Diagnostic build only: per-wave cycle stamps of the GEMM kernels (never compiled into the
// shipped library).  32 x 64-bit slots per wave.
__device__ unsigned long long g_stamps[4096 * 32];
#define STAMP_OK ((size_t)blockIdx.x * NWAVES + w < 4096)  /* larger grids: only the first 4096 waves are stamped */
#define STAMP(slot)                                                                      \
  do {                                                                                   \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                          \
    if (lane == 0 && STAMP_OK) g_stamps[((size_t)blockIdx.x * NWAVES + w) * 32 + (slot)] = t_; \
  } while (0)
#else
#define STAMP(slot) do {} while (0)
#endif

// MT : 16-row tiles of the output per workgroup
// T  : 16-column tiles per workgroup
// W  : wavefronts per workgroup (they split K)
// SZS: k-steps (of 32) covered by one scale/zero word: 4 when G % 128 == 0,
//      2 when G == 64, 1 for any other multiple of 32
// ACT: the weight is [gate | up] (two halves of N/2 columns, N/2 a multiple of 16); the workgroup
//      takes T/2 column tiles of each half and stores silu(gate) * up, an [M, N/2] output.
template <int MT, int T, int W, int SZS, bool ACT = false>
__global__ void __launch_bounds__(64 * W) gemm_packed_kernel(const GemmArgs ga) {
  fetch_args_now(ga);
  // a masked-out problem of a batched launch: the whole workgroup leaves before it touches memory
  if (ga.mask != nullptr && ga.mask[blockIdx.z] == 0) return;
  const GemmArgs a = batch_slice(ga);
  constexpr int NSZ = 4 / SZS;  // scale/zero words per 128-k block and tile
  constexpr int TM = T * MT;
  constexpr int H = ACT ? T / 2 : T;       // tiles per half
  static_assert(!ACT || T % 2 == 0, "gate and up tiles come in pairs");
  const int NTH = ACT ? a.NT / 2 : a.NT;   // column tiles per half
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw);

  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n16 = lane & 15;
  const int kq = lane >> 4;
  const int nt0 = blockIdx.x * H;
  const int m0 = blockIdx.y * (MT * 16);
#ifdef W4A16_STAMPS
  constexpr int NWAVES = W;
  int n_done = 0;
#endif
  STAMP(0);

  const u32x4* wp[T];
  const uint32_t* szp[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    // Tiles past the end re-read the last tile (valid memory); their results are dropped.
    const int nt = min(nt0 + (t % H), NTH - 1) + (t / H) * NTH;
    wp[t] = a.pw + (size_t)nt * a.KBS * 64 + lane;
    szp[t] = a.psz + (size_t)nt * a.NG * 16 + n16;
  }
  // Rows past M re-read row M-1: output row m depends on x row m only, and rows >= M
  // are never stored, so no masking is needed.
  const f16* xp[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = min(m0 + mt * 16 + n16, a.M - 1);
    xp[mt] = a.x + (size_t)row * a.K + kq * 8;
  }

  f32x4 acc[T][MT];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const DqConst dqc = make_dq_const();

  // For small row counts the x fragments of a whole block (4 k-steps) travel with the block's
  // weights: every load of a block is then issued up front, oldest-needed first (x from L2, the
  // scale/zero words, then the weight stream from HBM), and nothing is loaded inside the
  // compute phase.  For larger MT that would cost 16 * MT registers per buffer, so x fragments
  // are fetched one k-step ahead instead.
  constexpr bool XPRE = (MT <= 2);
  struct Block {
    f16x8 x[XPRE ? 4 : 1][MT];
    uint32_t sz[T][NSZ];
    u32x4 q[T];
  };

  // k-steps past K (K not a multiple of 128, or the prefetch after the last block)
  // re-read k = 0: the packer fills k >= K with q = z, i.e. exact zero weights.
  auto load_x = [&](int k, f16x8 (&xf)[MT]) {
    const int ke = (k < a.K) ? k : 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xf[mt] = *reinterpret_cast<const f16x8*>(xp[mt] + ke);
  };

  auto load_block = [&](int kb, Block& b) {
    if constexpr (XPRE) {
#pragma unroll
      for (int j = 0; j < 4; ++j) load_x(kb * 128 + j * 32, b.x[j]);
    }
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int s = 0; s < NSZ; ++s) {
        const int g = min(group_of<SZS>(a, kb * 128 + s * SZS * 32), a.NG - 1);
        b.sz[t][s] = szp[t][g * 16];
      }
#pragma unroll
    for (int t = 0; t < T; ++t) b.q[t] = __builtin_nontemporal_load(wp[t] + (size_t)kb * 64);
    // The scheduler must not sink these loads into the compute phase that follows.
    __builtin_amdgcn_sched_barrier(0);
  };

  // One 128-k block: 4 k-steps of (T dequantised A fragments) x (MT x fragments).
  // Without XPRE, `xf` holds the x fragments of this block's first k-step on entry and those of
  // block `kb_next`'s first k-step on exit (loaded one k-step ahead throughout).
  auto compute_block = [&](int kb, int kb_next, const Block& b, f16x8 (&xf)[MT]) {
    if constexpr (XPRE && W4A16_TMAJOR) {
      // Tile by tile, in the order the weight loads were issued: the wave starts on its first tile while
      // the others are still on their way (a wave's loads complete in issue order, so the compiler's wait
      // before tile t lets the loads of tiles t + 1.. stay outstanding).
#pragma unroll
      for (int t = 0; t < T; ++t) {
        SZ ct;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j % SZS == 0) ct = expand_sz(b.sz[t][j / SZS]);
          const f16x8 af = (W4A16_DBG & 8) ? __builtin_bit_cast(f16x8, b.q[t]) : dequant_word(b.q[t][j], ct, dqc);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, b.x[j][mt], acc[t][mt], 0, 0, 0);
        }
      }
      return;
    }
    SZ c[T];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f16x8 xn[MT];
      if constexpr (!XPRE) {
        const int knext = (j < 3) ? kb * 128 + (j + 1) * 32 : kb_next * 128;
        load_x(knext, xn);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (j % SZS == 0) {
#pragma unroll
        for (int t = 0; t < T; ++t) c[t] = expand_sz(b.sz[t][j / SZS]);
      }
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const f16x8 af = (W4A16_DBG & 8) ? __builtin_bit_cast(f16x8, b.q[t]) : dequant_word(b.q[t][j], c[t], dqc);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if constexpr (XPRE)
            acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, b.x[j][mt], acc[t][mt], 0, 0, 0);
          else
            acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, xf[mt], acc[t][mt], 0, 0, 0);
        }
      }
      if constexpr (!XPRE) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xf[mt] = xn[mt];
      }
    }
  };

  // K loop.  Blocks past KB are never computed; `kb_next * 128 >= K` turns the x prefetch past the
  // end into a re-read of k = 0.  Small tiles (one or two MFMAs per weight fragment) keep three
  // blocks in flight ahead of the one being computed instead of one: a wave's block is only
  // T KiB of weights, and with a long K (11008) one block ahead leaves the CU with too few bytes
  // in flight to cover HBM latency.
  if constexpr (XPRE && T * MT <= W4A16_DEEP_TM) {
    Block b0, b1, b2, b3;
    f16x8 xf[MT];
    int kb = w;
    if (kb < a.KB) load_block(kb, b0);
    if (kb + W < a.KB) load_block(kb + W, b1);
    if (kb + 2 * W < a.KB) load_block(kb + 2 * W, b2);
    while (kb < a.KB) {
      if (kb + 3 * W < a.KB) load_block(kb + 3 * W, b3);
      compute_block(kb, kb + W, b0, xf);
      kb += W;
      if (kb >= a.KB) break;
      if (kb + 3 * W < a.KB) load_block(kb + 3 * W, b0);
      compute_block(kb, kb + W, b1, xf);
      kb += W;
      if (kb >= a.KB) break;
      if (kb + 3 * W < a.KB) load_block(kb + 3 * W, b1);
      compute_block(kb, kb + W, b2, xf);
      kb += W;
      if (kb >= a.KB) break;
      if (kb + 3 * W < a.KB) load_block(kb + 3 * W, b2);
      compute_block(kb, kb + W, b3, xf);
      kb += W;
    }
  } else {
    Block b0, b1;
    f16x8 xf[MT];
    int kb = w;
    if (kb < a.KB) {
      if constexpr (!XPRE) load_x(kb * 128, xf);
      load_block(kb, b0);
    }
    STAMP(1);
    while (kb < a.KB) {
      int kn = kb + W;
      if (kn < a.KB) load_block(kn, b1);
      compute_block(kb, kn, b0, xf);
#ifdef W4A16_STAMPS
      STAMP(2 + min(n_done, 7)); ++n_done;
#endif
      kb = kn;
      if (kb >= a.KB) break;
      kn = kb + W;
      if (kn < a.KB) load_block(kn, b0);
      compute_block(kb, kn, b1, xf);
#ifdef W4A16_STAMPS
      STAMP(2 + min(n_done, 7)); ++n_done;
#endif
      kb = kn;
    }
  }
  STAMP(10);

  // Cross-wave reduction through LDS: fold the upper half of the live waves onto the
  // lower half until LIVE partial sets fit the budget, then every wave sums whole tiles.
  constexpr int LIVE = live_waves(W, TM);
  // C layout of the MFMA: register r of lane l is row 4 (l >> 4) + r of the tile (a column n of
  // `out`), column l & 15 (a row m of `out`).  Values are rounded to fp16 first, then the bias is
  // added in fp16: the value of `out + bias` on an fp16 GEMM result (awq.py:163-164).
  auto finish = [&](f32x4 v, int n_bias) {
    f16x4 h = f16x4{(f16)v.x, (f16)v.y, (f16)v.z, (f16)v.w};
    if (a.bias != nullptr) h += *reinterpret_cast<const f16x4*>(a.bias + n_bias);
    return h;
  };
  // i indexes (t, mt) pairs of the first H tiles; with ACT `vu` is the matching tile of the up half
  auto store_tile = [&](int i, f32x4 v, f32x4 vu) {
    const int t = i / MT, mt = i - t * MT;
    const int nt = nt0 + t;
    const int n = nt * 16 + kq * 4;
    const int m = m0 + mt * 16 + n16;
    if constexpr (ACT) {
      const int NH = a.N / 2;
      if (nt < NTH && m < a.M)
        *reinterpret_cast<f16x4*>(a.out + (size_t)m * NH + n) = silu_mul(finish(v, n), finish(vu, NH + n));
    } else {
      if (nt < a.NT && n < a.N && m < a.M) {
        f16x4* dst = reinterpret_cast<f16x4*>(a.out + (size_t)m * a.N + n);
        f16x4 h = finish(v, n);
        // fp16 + fp16, as `residual + linear(x)` on fp16 tensors; one lane owns each element
        if (a.add_out) h = *dst + h;
        *dst = h;
      }
    }
  };
  constexpr int OUTS = H * MT;  // stores per workgroup: tiles, or gate/up tile pairs
  if (W > 1 && MT == 1 && W4A16_SLIM && a.M <= 4) {
    // Up to 4 rows: of a wave's 16 x 16 partial tile only the lanes of tokens 0..M-1 (l & 15 < M) hold
    // anything that is stored.  They park their 16 bytes compactly, [wave][tile][token][l >> 4], 256 bytes
    // per wave and tile instead of 1 KiB, and the first wave alone sums the W partials of every
    // (tile, token, 4-column group): W independent 16-byte reads per lane, one round trip.
    if (n16 < 4) {
#pragma unroll
      for (int t = 0; t < T; ++t) red[((w * T + t) * 4 + n16) * 4 + kq] = acc[t][0];
    }
    __syncthreads();
    if (w != 0) return;
    // item = (tile or tile pair, token, column group): 16 per tile, 64 per pass of the wave
    for (int item = lane; item < OUTS * 16; item += 64) {
      const int ti = item / 16, tok = (item >> 2) & 3, grp = item & 3;
      if (tok >= a.M) continue;
      f32x4 v = red[(ti * 4 + tok) * 4 + grp];
#pragma unroll
      for (int s = 1; s < W; ++s) v += red[((s * T + ti) * 4 + tok) * 4 + grp];
      f32x4 vu = v;
      if constexpr (ACT) {
        vu = red[((ti + H) * 4 + tok) * 4 + grp];
#pragma unroll
        for (int s = 1; s < W; ++s) vu += red[((s * T + ti + H) * 4 + tok) * 4 + grp];
      }
      const int nt = nt0 + ti, n = nt * 16 + grp * 4, m = m0 + tok;
      if constexpr (ACT) {
        const int NH = a.N / 2;
        if (nt < NTH) *reinterpret_cast<f16x4*>(a.out + (size_t)m * NH + n) = silu_mul(finish(v, n), finish(vu, NH + n));
      } else {
        if (nt < a.NT && n < a.N) {
          f16x4* dst = reinterpret_cast<f16x4*>(a.out + (size_t)m * a.N + n);
          f16x4 hv = finish(v, n);
          if (a.add_out) hv = *dst + hv;
          *dst = hv;
        }
      }
    }
    return;
  }
  if constexpr (W == 1) {
#pragma unroll
    for (int i = 0; i < OUTS; ++i)
      store_tile(i, acc[i / MT][i % MT], acc[(i / MT + (ACT ? H : 0)) % T][i % MT]);
  } else {
#pragma unroll
    for (int cur = W; cur > LIVE; cur >>= 1) {
      const int half = cur >> 1;
      if (w >= half && w < cur) {
#pragma unroll
        for (int i = 0; i < TM; ++i) red[((w - half) * TM + i) * 64 + lane] = acc[i / MT][i % MT];
      }
      __syncthreads();
      if (w < half) {
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i / MT][i % MT] += red[(w * TM + i) * 64 + lane];
      }
      __syncthreads();
    }
    if (w < LIVE) {
#pragma unroll
      for (int i = 0; i < TM; ++i) red[(w * TM + i) * 64 + lane] = acc[i / MT][i % MT];
    }
    __syncthreads();
    STAMP(11);
    for (int i = w; i < OUTS; i += W) {
      f32x4 v = red[i * 64 + lane];
#pragma unroll
      for (int s = 1; s < LIVE; ++s) v += red[(s * TM + i) * 64 + lane];
      f32x4 vu = v;
      if constexpr (ACT) {
        const int iu = i + OUTS;
        vu = red[iu * 64 + lane];
#pragma unroll
        for (int s = 1; s < LIVE; ++s) vu += red[(s * TM + iu) * 64 + lane];
      }
      store_tile(i, v, vu);
    }
  }
  STAMP(12);
}

// A compile-time switch passed as a lambda argument.
template <int V>
struct Flag {
  static constexpr int value = V;
};

// Waits until at most N of the wave's vector-memory operations (which complete in issue order) are outstanding.
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---------------------------------------------------------------------------
// Decode GEMM on the private layout: up to 16 rows, paced weight stream, optional K split over workgroups
// ---------------------------------------------------------------------------
// The streaming kernel above requests everything a wave will ever need at once, activations included, in
// fragment-shaped loads: at one row a wave of the 4096 x 11008 launch issues 20 vector-memory instructions for
// 6 KiB of weights.  A CU holds only so many bytes in flight (about 50-60 KiB) and admits requests oldest wave
// first, so the youngest waves of a workgroup get theirs in thousands of cycles after the oldest, receive their
// data last and still have all of their arithmetic to do after the last byte.
//
// Here a wave's activations (its k-blocks x up to 4 rows x 256 bytes) go to a private piece of LDS once, ahead
// of the weights, by one LDS-DMA instruction per block (rows past M - 1 repeat the last row), and come back as
// ds_read_b128 B fragments: chunk c of activation row r sits at 16-byte position c ^ 4r of the row's 256
// bytes, so the rows a fragment read touches fall into different banks.  The weights come in units of 1 KiB
// (one tile's 128-k block, one instruction, with the tile's scale/zero word(s) for the block); a wave keeps DB
// blocks of T units in flight and asks for the unit DB blocks ahead as soon as it has consumed a unit.
//
// gridDim.y > 1 splits K over workgroups: each writes its fp32 partial sums (after the usual reduction over
// its own waves) to its slot of `ws`, takes a ticket, and the workgroup that draws the last ticket of an
// output slot adds the slots in split order (so the result does not depend on who came last), applies the
// epilogue, stores, and returns the ticket counter to zero for the next launch.
// Launch constants of the decode kernel the host works out once, so that the 16 waves of a workgroup, which
// share one scalar unit, do not each derive them again before their first load can go out.
struct DecodeArgs {
  f32x4* ws;          // K-split launches: partial sums
  unsigned* tickets;  // K-split launches: arrival counters
  int kb_per_split;   // 128-k blocks per workgroup
  int nb_max;         // most blocks any wave takes
  int wstride;        // bytes of one column tile in packed_w
  int sstride;        // bytes of one column tile in packed_sz
  int steal;          // the oldest quarter of the waves takes one block each from the youngest quarter
  int splits;         // workgroups per output slot (gridDim.y; here so that nobody fetches it at the very end)
  int gx;             // workgroup columns (gridDim.x)
};

// XLDS = false serves 5-16 rows: the activations then come as B fragments straight from L2, one block ahead of
// the block being worked on (at 16 rows a fragment load has no lane to spare, and 16 rows x 256 bytes per block
// and wave would not fit LDS), the waves park whole 16 x 16 partial tiles, and waves 0..H-1 each sum one tile.
template <int T, int W, int DB, int SZS, bool ACT = false, bool XLDS = true>
__global__ void __launch_bounds__(64 * W) gemm_decode_kernel(const GemmArgs ga, const DecodeArgs e) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub (and has no buffer-resource type)
  fetch_args_now(ga);
  asm volatile("" ::"s"(e.ws), "s"(e.tickets), "s"(e.kb_per_split), "s"(e.nb_max), "s"(e.wstride), "s"(e.sstride),
               "s"(e.steal), "s"(e.splits), "s"(e.gx));
  if (ga.mask != nullptr && ga.mask[blockIdx.z] == 0) return;
  const GemmArgs a = batch_slice(ga);
  constexpr int NSZ = 4 / SZS;             // scale/zero words per 128-k block and tile
  constexpr int UL = NSZ + 1;              // vector-memory loads per unit
  constexpr int H = ACT ? T / 2 : T;       // tiles per half
  static_assert(DB == 1 || DB == 2, "one or two blocks in flight");
  static_assert(!ACT || T % 2 == 0, "gate and up tiles come in pairs");
  static_assert(H <= 8, "one wave finishes all tiles: 16 lanes per tile or tile pair, in one or two passes");
  const int NTH = ACT ? a.NT / 2 : a.NT;   // column tiles per half
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int nb_max = e.nb_max;

  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n16 = lane & 15;
  const int kq = lane >> 4;
  const int nt0 = blockIdx.x * H;
#ifdef W4A16_STAMPS
  constexpr int NWAVES = W;
  int n_done = 0;
#endif
  STAMP(0);
  // this workgroup's k-blocks [kb_lo, kb_hi); this wave's: kb0, kb0 + W, ..
  const int kb_lo = blockIdx.y * e.kb_per_split;
  const int kb_hi = min(kb_lo + e.kb_per_split, a.KB);
  const int kb0 = kb_lo + w;
  const int own = kb0 < kb_hi ? (kb_hi - kb0 + W - 1) / W : 0;  // blocks kb0, kb0 + W, ..
  // Requests are admitted oldest wave first, so the youngest quarter of the waves gets its data last and would
  // finish last.  Where every wave has at least two blocks, each wave of that quarter hands its last block to
  // the wave three quarters of a workgroup older, which has long been waiting by then.
  int nblk = own, kb_extra = 0;
  if (W4A16_DECODE_STEAL && e.steal) {
    constexpr int Q = W / 4;
    if (w >= 3 * Q) {
      nblk = own - 1;
    } else if (w < Q) {
      const int young = kb0 + 3 * Q;  // first block of the wave this one takes a block from
      const int young_own = young < kb_hi ? (kb_hi - young + W - 1) / W : 0;
      kb_extra = young + (young_own - 1) * W;
      nblk = own + 1;
    }
  }
  auto kb_of = [&](int bi) { return bi < own ? kb0 + bi * W : kb_extra; };

  const __amdgpu_buffer_rsrc_t w_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(a.pw), 0, a.NT * e.wstride, 0x00020000);
  const __amdgpu_buffer_rsrc_t sz_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.psz), 0, a.NT * e.sstride, 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(a.x), 0, a.M * a.K * 2, 0x00020000);
  // Uniform byte offsets of the workgroup's tiles (tiles past the end re-read the last one; their results are
  // dropped); the per-lane parts are the same for every tile and block.
  int wbase[T], szbase[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int nt = min(nt0 + (t % H), NTH - 1) + (t / H) * NTH;
    wbase[t] = nt * e.wstride;
    szbase[t] = nt * e.sstride;
  }
  const uint32_t wlane = lane * 16u, szlane = n16 * 4u;
  // LDS-DMA of one block's activations (1 KiB of the wave's private LDS): lane -> row lane >> 4 (rows past
  // M - 1 repeat it), 16-byte position lane & 15 of the row, which holds chunk (lane & 15) ^ 4 (lane >> 4)
  const uint32_t xlane = ((uint32_t)min(kq, a.M - 1) * a.K + ((n16 ^ (4 * kq)) * 8)) * 2u;
  const int xwave = w * nb_max * 1024;
  // B fragment of k-step j: row min(n16, M - 1) (the matrix instruction's columns past M - 1 are never stored),
  // chunk 4 j + kq
  int xrd[4];
  {
    const int rr = min(n16, a.M - 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) xrd[j] = xwave + rr * 256 + ((4 * (j ^ rr) + kq) << 4);
  }

  f32x4 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const DqConst dqc = make_dq_const();

  struct Unit {
    uint32_t sz[NSZ];
    u32x4 q;
  };
  auto load_unit = [&](int t, int kb, Unit& d) {
#pragma unroll
    for (int s = 0; s < NSZ; ++s) {
      // the host admits group sizes of 32, 64 and multiples of 128 only
      const int g = SZS == 4 ? (kb * a.g_magic) >> 16 : kb * NSZ + s;
      d.sz[s] = __builtin_amdgcn_raw_buffer_load_b32(sz_rs, szlane, szbase[t] + g * 64, 0);
    }
    d.q = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rs, wlane, wbase[t] + kb * 1024, 2));  // nt
  };
  auto compute_unit = [&](int t, const Unit& d, const f16x8 (&xf)[4]) {
    SZ c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j % SZS == 0) c = expand_sz(d.sz[j / SZS]);
      const f16x8 af = dequant_word(d.q[j], c, dqc);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, xf[j], acc[t], 0, 0, 0);
    }
  };
  // Without LDS staging: the B fragments of a block (row min(n16, M - 1), k = 32 j + 8 kq .. + 7), 4 loads
  const uint32_t xfrag_lane = ((uint32_t)min(n16, a.M - 1) * a.K + kq * 8) * 2u;
  auto load_x = [&](int kb, f16x8 (&xf)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      xf[j] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(x_rs, xfrag_lane, (kb * 128 + j * 32) * 2, 0));
  };
  // One block from the registers `u`; with REFILL each unit's registers are asked to fetch the same tile's unit
  // of block kb_refill as soon as the unit is consumed; with XNEXT (fragments from L2 only) the next block's
  // fragments are requested into `xnext` first.  (Compile-time flags: a load under a run-time branch makes the
  // compiler wait for every load in flight.)
  auto step = [&](auto refill_, auto xnext_, int bi, Unit (&u)[T], f16x8 (&xcur)[4], f16x8 (&xnext)[4], int kb_refill,
                  int kb_xnext) {
    constexpr bool REFILL = decltype(refill_)::value != 0;
    constexpr bool XNEXT = decltype(xnext_)::value != 0;
    f16x8 xl[4];
    if constexpr (XLDS) {
#pragma unroll
      for (int j = 0; j < 4; ++j) xl[j] = *reinterpret_cast<const f16x8*>(smem_raw + xrd[j] + bi * 1024);
    } else if constexpr (XNEXT) {
      load_x(kb_xnext, xnext);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if constexpr (XLDS)
        compute_unit(t, u[t], xl);
      else
        compute_unit(t, u[t], xcur);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (REFILL) {
        load_unit(t, kb_refill, u[t]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#ifdef W4A16_STAMPS
    STAMP(2 + min(n_done, 7)); ++n_done;
#endif
  };

  if (nblk > 0) {
    f16x8 xa[4], xb[4];
    if constexpr (XLDS) {
      // activations first: they come from L2 and are back before the weights
      for (int bi = 0; bi < nblk; ++bi)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr)(smem_raw + xwave + bi * 1024), 16, xlane,
                                                 kb_of(bi) * 256, 0, 0);
    } else {
      load_x(kb0, xa);
    }
    __builtin_amdgcn_sched_barrier(0);
    Unit u0[T], u1[T];
#pragma unroll
    for (int t = 0; t < T; ++t) load_unit(t, kb0, u0[t]);
    if constexpr (DB == 2) {
      // (a wave with a single block fetches it twice: nothing in the prologue is conditional)
      const int kb1 = nblk > 1 ? kb_of(1) : kb0;
#pragma unroll
      for (int t = 0; t < T; ++t) load_unit(t, kb1, u1[t]);
    }
    __builtin_amdgcn_sched_barrier(0);
    STAMP(1);
    // everything but the DB * T units just requested has landed: the activations are in LDS
    // (the counter has 6 bits: with more than 63 loads behind the DMA a few of the weights are waited for too)
    if constexpr (XLDS) wait_vmcnt<(DB * T * UL < 63 ? DB * T * UL : 63)>();
    constexpr Flag<1> Y{};
    constexpr Flag<0> N{};
    // Even blocks read fragment set xa, odd blocks xb (LDS staging ignores both).
    if constexpr (DB == 1) {
      int bi = 0;
      for (; bi + 2 < nblk; bi += 2) {
        step(Y, Y, bi, u0, xa, xb, kb_of(bi + 1), kb_of(bi + 1));
        step(Y, Y, bi + 1, u0, xb, xa, kb_of(bi + 2), kb_of(bi + 2));
      }
      if (nblk - bi == 2) {
        step(Y, Y, bi, u0, xa, xb, kb_of(bi + 1), kb_of(bi + 1));
        step(N, N, bi + 1, u0, xb, xa, 0, 0);
      } else {
        step(N, N, bi, u0, xa, xb, 0, 0);
      }
    } else {
      // u0 holds block bi, u1 block bi + 1; blocks that have a block two ahead go two per trip
      int bi = 0;
      for (; bi + 4 <= nblk; bi += 2) {
        step(Y, Y, bi, u0, xa, xb, kb_of(bi + 2), kb_of(bi + 1));
        step(Y, Y, bi + 1, u1, xb, xa, kb_of(bi + 3), kb_of(bi + 2));
      }
      const int rest = nblk - bi;
      if (rest == 3) {
        step(Y, Y, bi, u0, xa, xb, kb_of(bi + 2), kb_of(bi + 1));
        step(N, Y, bi + 1, u1, xb, xa, 0, kb_of(bi + 2));
        step(N, N, bi + 2, u0, xa, xb, 0, 0);
      } else if (rest == 2) {
        step(N, Y, bi, u0, xa, xb, 0, kb_of(bi + 1));
        step(N, N, bi + 1, u1, xb, xa, 0, 0);
      } else {
        step(N, N, bi, u0, xa, xb, 0, 0);
      }
    }
  }
  STAMP(10);

  // Reduction over the waves: of a 16 x 16 partial tile only the lanes of tokens 0..3 (lane & 15 < 4) hold
  // anything that is stored; they park their 16 bytes as [wave][tile][token][lane >> 4] and the first wave
  // sums the W partials of every (tile, token, 4-column group), one lane each.
  if constexpr (!XLDS) {
    // 5-16 rows: whole partial tiles (1 KiB per wave and tile), wave ti sums tile ti (and its up-half twin)
    f32x4* red = reinterpret_cast<f32x4*>(smem_raw);
#pragma unroll
    for (int t = 0; t < T; ++t) red[(w * T + t) * 64 + lane] = acc[t];
    __syncthreads();
    STAMP(11);
    if (w >= H) return;
    const int ti = w;
    f32x4 v = red[ti * 64 + lane], vu = v;
#pragma unroll
    for (int s = 1; s < W; ++s) v += red[(s * T + ti) * 64 + lane];
    if constexpr (ACT) {
      vu = red[(ti + H) * 64 + lane];
#pragma unroll
      for (int s = 1; s < W; ++s) vu += red[(s * T + ti + H) * 64 + lane];
    }
    auto finish = [&](f32x4 r, int n_bias) {
      f16x4 h = f16x4{(f16)r.x, (f16)r.y, (f16)r.z, (f16)r.w};
      if (a.bias != nullptr) h += *reinterpret_cast<const f16x4*>(a.bias + n_bias);
      return h;
    };
    const int nt = nt0 + ti, n = nt * 16 + kq * 4, tok = n16;
    if (tok >= a.M) return;
    if constexpr (ACT) {
      const int NH = a.N / 2;
      if (nt < NTH) *reinterpret_cast<f16x4*>(a.out + (size_t)tok * NH + n) = silu_mul(finish(v, n), finish(vu, NH + n));
    } else {
      if (nt < a.NT && n < a.N) {
        f16x4* dst = reinterpret_cast<f16x4*>(a.out + (size_t)tok * a.N + n);
        f16x4 hv = finish(v, n);
        if (a.add_out) hv = *dst + hv;
        *dst = hv;
      }
    }
    STAMP(12);
    return;
  }
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw + W * nb_max * 1024);
  f32x4* const ws = e.ws;
  unsigned* const tickets = e.tickets;
  if (n16 < 4) {
#pragma unroll
    for (int t = 0; t < T; ++t) red[((w * T + t) * 4 + n16) * 4 + kq] = acc[t];
  }
  __syncthreads();
  STAMP(11);
  if (w != 0) return;
  constexpr int PASSES = (H * 16 + 63) / 64;  // 16 lanes per tile (pair): one pass up to 4, two up to 8
#pragma unroll
  for (int pass = 0; pass < PASSES; ++pass) {
  const int ti = pass * 4 + lane / 16, tok = (lane >> 2) & 3, grp = lane & 3;
  const bool live = ti < H && tok < a.M;
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f}, vu = v;
  if (live) {
    v = red[(ti * 4 + tok) * 4 + grp];
#pragma unroll
    for (int s = 1; s < W; ++s) v += red[((s * T + ti) * 4 + tok) * 4 + grp];
    if constexpr (ACT) {
      vu = red[((ti + H) * 4 + tok) * 4 + grp];
#pragma unroll
      for (int s = 1; s < W; ++s) vu += red[((s * T + ti + H) * 4 + tok) * 4 + grp];
    }
  }
  const int S = e.splits;
  if (PASSES == 1 && S > 1) {  // (the host splits K only for launch shapes of up to 4 tiles or tile pairs)
    constexpr int SLOT = ACT ? 128 : 64;  // f32x4 per workgroup
    const size_t slot = (size_t)blockIdx.z * e.gx + blockIdx.x;
    // Partial sums and tickets travel as device-scope atomic accesses (each one coherent across the XCDs'
    // L2s by itself), ordered by the wave's own completion counter: a device-scope fence would write back and
    // invalidate a whole L2 per workgroup (~50 ns each, serialised: 69 us on a 1376-workgroup launch).
    float* all = reinterpret_cast<float*>(ws + slot * S * SLOT);
    auto put = [&](int idx, f32x4 r) {
#pragma unroll
      for (int e = 0; e < 4; ++e) __hip_atomic_store(all + idx * 4 + e, r[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto get = [&](int idx) {
      f32x4 r;
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = __hip_atomic_load(all + idx * 4 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return r;
    };
    put(blockIdx.y * SLOT + lane, v);
    if constexpr (ACT) put(blockIdx.y * SLOT + 64 + lane, vu);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the partial sums have arrived before the ticket is drawn
    unsigned old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(&tickets[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != (unsigned)(S - 1)) return;  // whole wave
    // last of the S workgroups of this output slot: the others' sums were complete before their tickets
    v = get(lane);
    for (int s = 1; s < S; ++s) v += get(s * SLOT + lane);
    if constexpr (ACT) {
      vu = get(64 + lane);
      for (int s = 1; s < S; ++s) vu += get(s * SLOT + 64 + lane);
    }
    if (lane == 0) __hip_atomic_store(&tickets[slot], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // as the next launch expects it
  }
  if (!live) continue;
  auto finish = [&](f32x4 r, int n_bias) {
    f16x4 h = f16x4{(f16)r.x, (f16)r.y, (f16)r.z, (f16)r.w};
    if (a.bias != nullptr) h += *reinterpret_cast<const f16x4*>(a.bias + n_bias);
    return h;
  };
  const int nt = nt0 + ti, n = nt * 16 + grp * 4;
  if constexpr (ACT) {
    const int NH = a.N / 2;
    if (nt < NTH) *reinterpret_cast<f16x4*>(a.out + (size_t)tok * NH + n) = silu_mul(finish(v, n), finish(vu, NH + n));
  } else {
    if (nt < a.NT && n < a.N) {
      f16x4* dst = reinterpret_cast<f16x4*>(a.out + (size_t)tok * a.N + n);
      f16x4 hv = finish(v, n);
      if (a.add_out) hv = *dst + hv;
      *dst = hv;
    }
  }
  }  // pass
  STAMP(12);
#endif
}

// ---------------------------------------------------------------------------
// GEMM on the private layout, activations staged through LDS (M > 32)

// Workgroup id -> (column slice, row block) for the kernels whose workgroups own a [rows] x [columns] tile.
// The work items are listed panel by panel (a panel = `panel_rows` row blocks: all of them up to 8, else a few),
// within a panel slice by slice with the panel's row blocks of a slice next to each other.  Workgroup ids
// congruent to x mod 8 run on XCD x, which takes the x-th eighth of that list in order: the workgroups an XCD
// runs at a time then share their weight slices AND their activation rows through its L2, and every XCD gets
// the same number of items whatever the slice count is (slices dealt out modulo 8 left up to a whole slice's
// row blocks more on some XCDs: 36 workgroups against 24 on 32 CUs in bad cases).
__device__ __forceinline__ bool map_workgroup(int n_slices, int m_blocks, int panel_rows, int mb_magic, int& slice,
                                              int& mb) {
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int total = n_slices * m_blocks;
  const int per = (total + 7) >> 3;
  const int j = xcd * per + idx;
  if (j >= total) return false;
  if (mb_magic != 0) {  // one panel of m_blocks row blocks: no division
    slice = (j * mb_magic) >> 16;
    mb = j - slice * m_blocks;
    return true;
  }
  const int per_panel = panel_rows * n_slices;
  const int panel = j / per_panel, in_panel = j - panel * per_panel;
  const int rows_here = min(panel_rows, m_blocks - panel * panel_rows);
  slice = in_panel / rows_here;
  mb = panel * panel_rows + in_panel % rows_here;
  return true;
}

// ---------------------------------------------------------------------------
// A workgroup owns MT x (NW * T) output tiles over the whole of K.  Its NW * KW waves form
// KW K-groups (group kg takes 128-k blocks kg, kg + KW, ...) of NW waves that split the
// workgroup's columns.  The [MT*16 rows] x [128 k] activation slab of a K-group's block is
// fetched ONCE per workgroup with full-line coalesced loads (each wave a share), parked in LDS
// (XOR-swizzled 256-byte rows, double buffered) and read back as MFMA B fragments by the NW
// waves of the group; weights still go HBM -> VGPR -> matrix core with no LDS round trip.
// Against the streaming kernel this removes the fragment-shaped global loads of x (16 rows x
// 64 B per instruction, every 128-byte line fetched twice) that bound it at M >= 64.
//
// Load order per block iteration: next block's x share, then the weights two blocks ahead.
// Vector-memory returns are counted in order, so this gives the L2-resident x one block and
// the HBM weight stream two blocks of cover.

// ACT (gate | up weight, silu(gate) * up stored, see gemm_packed_kernel): the first NW / 2 waves of a
// K-group take the workgroup's gate columns, the others the matching up columns.
template <int MT, int T, int NW, int KW, int SZS, bool ACT = false>
__global__ void __launch_bounds__(64 * NW * KW) gemm_tiled_kernel(int n_slices, int m_blocks, int panel_rows,
                                                                   const GemmArgs ga) {
  fetch_args_now(ga);
  asm volatile("" ::"s"(n_slices), "s"(m_blocks), "s"(panel_rows));
  // a masked-out problem of a batched launch: the whole workgroup leaves before it touches memory
  if (ga.mask != nullptr && ga.mask[blockIdx.z] == 0) return;
  const GemmArgs a = batch_slice(ga);
  constexpr int NWAVES = NW * KW;
  constexpr int HW = ACT ? NW / 2 : NW;    // waves per half
  static_assert(!ACT || NW % 2 == 0, "gate and up waves come in pairs");
  const int NTH = ACT ? a.NT / 2 : a.NT;   // column tiles per half
  constexpr int NSZ = 4 / SZS;
  constexpr int ROWS = MT * 16;
  constexpr int SLAB = ROWS * 256;                       // bytes: ROWS x 128 k of fp16
  constexpr int XLOADS = (ROWS / 4 + NW - 1) / NW;       // 16-byte loads per lane per slab share
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kg = w / NW, nw = w % NW;
  const int n16 = lane & 15;
  const int kq = lane >> 4;
  STAMP(0);
#ifdef W4A16_STAMPS
  unsigned long long t_work = 0, t_bar = 0;
#endif

  // Workgroups that share an XCD's L2 are those whose ids are equal mod 8.  Within an XCD the
  // (column slice, row-block) pairs are walked in panels of `panel_rows` row-blocks: slice after
  // slice, and for each slice the panel's row-blocks next to each other, so that the ~32 workgroups
  // an XCD runs at a time share their weight slices AND their activation rows through its L2.
  // Up to 8 row-blocks the panel is all of them: the row-blocks of a slice get ids 8 apart.
  int slice, mb;
  if (!map_workgroup(n_slices, m_blocks, panel_rows, a.mb_magic, slice, mb)) return;  // whole workgroup
  const int nt0 = (slice * HW + nw % HW) * T;
  const int m0 = mb * ROWS;

  const u32x4* wp[T];
  const uint32_t* szp[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int nt = min(nt0 + t, NTH - 1) + (nw / HW) * NTH;
    wp[t] = a.pw + (size_t)nt * a.KBS * 64 + lane;
    szp[t] = a.psz + (size_t)nt * a.NG * 16 + n16;
  }

  // This wave's share of a slab: pieces of 4 rows x 256 B; lane l -> row 4p + (l >> 4), chunk l & 15.
  const f16* xsrc[XLOADS];
  int xdst[XLOADS];
#pragma unroll
  for (int p = 0; p < XLOADS; ++p) {
    // (when NW does not divide the piece count the last wave repeats the last piece: same data
    // to the same place)
    const int piece = nw * XLOADS + p;
    const int row = ((ROWS / 4) % NW != 0 ? min(piece, ROWS / 4 - 1) : piece) * 4 + kq;
    const int grow = min(m0 + row, a.M - 1);
    xsrc[p] = a.x + (size_t)grow * a.K + n16 * 8;
    xdst[p] = kg * SLAB + row * 256 + ((n16 ^ (row & 15)) << 4);
  }
  // B fragment of row tile mt, k-step j: row 16 mt + n16, 16-byte chunk 4 j + kq of the row.
  int xfrag[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) xfrag[mt] = kg * SLAB + (mt * 16 + n16) * 256;

  f32x4 acc[T][MT];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const DqConst dqc = make_dq_const();

  struct WBlock {
    uint32_t sz[T][NSZ];
    u32x4 q[T];
  };
  auto load_w = [&](int kb, WBlock& b) {
    if ((W4A16_DBG & 2) && kb != kg) return;
    const int kbe = (W4A16_DBG & 128) ? kg : min(kb, a.KBS - 1);
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int s = 0; s < NSZ; ++s) {
        const int g = min(group_of<SZS>(a, kbe * 128 + s * SZS * 32), a.NG - 1);
        b.sz[t][s] = szp[t][g * 16];
      }
    // Default cache policy: the row-blocks of a column slice run side by side on one XCD and the
    // later ones are meant to find these lines in its L2 (a non-temporal load would drop them).
#pragma unroll
    for (int t = 0; t < T; ++t) b.q[t] = wp[t][(size_t)kbe * 64];
  };
  // loads pieces [p0, p1) of this wave's share of block kb's slab
  auto load_xshare = [&](int kb, u32x4 (&xr)[XLOADS], int p0, int p1) {
    // chunks past K (K % 128 != 0 or a block past the end) re-read k = 0: their weights are zero
    if ((W4A16_DBG & 1) && kb != kg) return;
    const int k = kb * 128 + n16 * 8;
    const int koff = (k < a.K) ? kb * 128 : -(n16 * 8);
#pragma unroll
    for (int p = p0; p < p1; ++p) xr[p] = *reinterpret_cast<const u32x4*>(xsrc[p] + koff);
  };
  auto store_xshare = [&](int buf, const u32x4 (&xr)[XLOADS]) {
#pragma unroll
    for (int p = 0; p < XLOADS; ++p)
      *reinterpret_cast<u32x4*>(smem_raw + buf * (KW * SLAB) + xdst[p]) = xr[p];
  };
  // k-steps [j0, j1) of a block; `c` carries the expanded scale/zero constants across calls.
  auto compute = [&](int buf, const WBlock& b, SZ (&c)[T], f16x8 (&xf)[MT], int j0, int j1) {
    const unsigned char* base = smem_raw + buf * (KW * SLAB);
#pragma unroll
    for (int j = j0; j < j1; ++j) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if ((W4A16_DBG & 16) && j > 0) continue;  // ablation: no LDS fragment reads after k-step 0
        xf[mt] = *reinterpret_cast<const f16x8*>(base + xfrag[mt] + (((4 * j + kq) ^ n16) << 4));
      }
      if (j % SZS == 0) {
#pragma unroll
        for (int t = 0; t < T; ++t) c[t] = expand_sz(b.sz[t][j / SZS]);
      }
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const f16x8 af = dequant_word(b.q[t][j], c[t], dqc);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, xf[mt], acc[t][mt], 0, 0, 0);
      }
    }
  };

  // All K-groups run KBS / KW iterations in lockstep (KBS is a multiple of 4 >= KW; blocks past K
  // hold zero weights), and nothing in the steady-state loop is conditional: a load issued under
  // a branch makes hipcc's wait-count bookkeeping pessimistic and drains the prefetches every
  // iteration.  The three weight buffers are named statically (a register that a load is still
  // writing cannot be renamed by a move), hence the loop is unrolled by 3 with a peeled tail.
  const int iters = (W4A16_DBG & 32) ? 0 : a.KBS / KW;
  WBlock w0, w1, w2;
  u32x4 xr[XLOADS];
  // A CU's vector-memory returns come back in issue order across all its waves, so an
  // L2-resident x load issued behind an HBM weight load waits for HBM, and a burst of loads at
  // the top of a phase keeps every wave in the issue queue instead of the matrix core.  The x
  // share is therefore issued in three portions ahead of k-steps 0-2 (all waves leave the barrier
  // together, so every wave's x loads precede every wave's weight loads), and the weight loads
  // follow k-step 2: they have until the end of the NEXT phase to land before they hold
  // anything up.
  auto phase = [&](int i, const WBlock& cur, WBlock& ahead) {
    const int kb = kg + i * KW;
    SZ c[T];
    f16x8 xf[MT];
#ifdef W4A16_STAMPS
    const unsigned long long ta = __builtin_amdgcn_s_memtime();
#endif
    // cumulative x pieces issued before k-step j (3/8, 3/8, 2/8 of the share), and the k-step
    // after which the weight loads follow
    constexpr int xs[4] = {3 * XLOADS / 8, 3 * XLOADS / 4, XLOADS, XLOADS};
    constexpr int wstep = 2;
    int done = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      load_xshare(kb + KW, xr, done, xs[j]);
      done = xs[j];
      __builtin_amdgcn_sched_barrier(0);
      compute(i & 1, cur, c, xf, j, j + 1);
      __builtin_amdgcn_sched_barrier(0);
      if (j == wstep) {
        load_w(kb + 2 * KW, ahead);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!(W4A16_DBG & 4)) store_xshare((i + 1) & 1, xr);  // ablation 4: no LDS stores
#ifdef W4A16_STAMPS
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): own LDS stores done, what the barrier needs anyway
    const unsigned long long tb = __builtin_amdgcn_s_memtime();
#endif
    if (!(W4A16_DBG & 8)) __syncthreads();  // ablation 8: no barrier
#ifdef W4A16_STAMPS
    const unsigned long long tc = __builtin_amdgcn_s_memtime();
    t_work += tb - ta;
    t_bar += tc - tb;
    if (lane == 0 && i < 8 && STAMP_OK) {
      g_stamps[((size_t)blockIdx.x * NWAVES + w) * 32 + 8 + 2 * i] = tb - ta;
      g_stamps[((size_t)blockIdx.x * NWAVES + w) * 32 + 9 + 2 * i] = tc - tb;
    }
#endif
  };
  // Prologue: only what the first block needs stands between launch and the first MFMA; the
  // second block's weights are requested once the first slab is on its way into LDS.
  load_xshare(kg, xr, 0, XLOADS);
  load_w(kg, w0);
  __builtin_amdgcn_sched_barrier(0);
  store_xshare(0, xr);
  __builtin_amdgcn_sched_barrier(0);
  load_w(kg + KW, w1);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  STAMP(1);
  int it = 0;
  for (; it + 3 <= iters; it += 3) {
    phase(it, w0, w2);
    phase(it + 1, w1, w0);
    phase(it + 2, w2, w1);
  }
  if (it < iters) {
    phase(it, w0, w2);
    if (it + 1 < iters) phase(it + 1, w1, w0);
  }

  // Reduce the KW partial sums of every output tile through LDS (the staging buffers are free
  // now: the loop ended on a barrier) and store.
  constexpr int TM = T * MT;
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw);
  STAMP(2);
#ifdef W4A16_STAMPS
  if (lane == 0 && STAMP_OK) {
    g_stamps[((size_t)blockIdx.x * NWAVES + w) * 32 + 5] = t_work;
    g_stamps[((size_t)blockIdx.x * NWAVES + w) * 32 + 6] = t_bar;
    g_stamps[((size_t)blockIdx.x * NWAVES + w) * 32 + 7] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  if (W4A16_DBG & 64) {
    if (acc[0][0][0] == 123.f) a.out[0] = (f16)1.0f;
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) red[((kg * NW + nw) * TM + i) * 64 + lane] = acc[i / MT][i % MT];
  __syncthreads();
  STAMP(3);
  // what happens to the finished fp32 tile `o` of this workgroup (`vu`: its up-half twin with ACT)
  auto finish_tile = [&](int o, f32x4 v, f32x4 vu) {
    const int onw = o / TM, i = o - onw * TM;
    const int t = i / MT, mt = i - t * MT;
    const int nt = (slice * HW + onw) * T + t;
    const int n = nt * 16 + kq * 4;
    const int m = m0 + mt * 16 + n16;
    f16x4 h = f16x4{(f16)v.x, (f16)v.y, (f16)v.z, (f16)v.w};
    if constexpr (ACT) {
      const int NH = a.N / 2;
      if (nt < NTH && m < a.M) {
        f16x4 hu = f16x4{(f16)vu.x, (f16)vu.y, (f16)vu.z, (f16)vu.w};
        if (a.bias != nullptr) {
          h += *reinterpret_cast<const f16x4*>(a.bias + n);
          hu += *reinterpret_cast<const f16x4*>(a.bias + NH + n);
        }
        *reinterpret_cast<f16x4*>(a.out + (size_t)m * NH + n) = silu_mul(h, hu);
      }
    } else {
      if (nt < a.NT && n < a.N && m < a.M) {
        if (a.bias != nullptr) h += *reinterpret_cast<const f16x4*>(a.bias + n);
        f16x4* dst = reinterpret_cast<f16x4*>(a.out + (size_t)m * a.N + n);
        if (a.add_out) h = *dst + h;  // fp16 + fp16; one lane owns each element
        *dst = h;
      }
    }
  };
  for (int o = w; o < HW * TM; o += NWAVES) {
    const int onw = o / TM, i = o - onw * TM;
    f32x4 v = red[(onw * TM + i) * 64 + lane];
#pragma unroll
    for (int g = 1; g < KW; ++g) v += red[((g * NW + onw) * TM + i) * 64 + lane];
    f32x4 vu = v;
    if constexpr (ACT) {
      vu = red[((HW + onw) * TM + i) * 64 + lane];
#pragma unroll
      for (int g = 1; g < KW; ++g) vu += red[((g * NW + HW + onw) * TM + i) * 64 + lane];
    }
    finish_tile(o, v, vu);
  }
  STAMP(4);
}

// ---------------------------------------------------------------------------
// GEMM on the private layout with 32x32x16 matrix instructions (M > 32)
// ---------------------------------------------------------------------------
// v_mfma_f32_32x32x16_f16 does twice the work of the 16x16x32 form per instruction and per converted
// weight word, which halves the int4 -> fp16 conversion's share of the vector issue slots.  Its A
// operand wants lane l to hold 8 k of weight column l & 31, k-octet l >> 5 of a 16-k step.  The packed
// layout (16-column tiles, lane quads of four 32-k words) serves it as it is: lanes with bit 4 clear
// read the even 16-column tile of the pair, the others the odd one, from source lanes
// (l & 15) + 16 (2 s + (l >> 5)), s = 0, 1: two 16-byte loads per lane and 128-k block, word j of load
// s being the fragment of k-step 2 j + s (k = 128 kb + 32 j + 16 s + 8 (l >> 5) + i).
//
// A workgroup owns MT x T tiles of 32 x 32 outputs over the whole of K; its KW waves split K (wave w
// takes blocks w, w + KW, ...) and each holds all MT x T accumulator tiles.  A wave keeps the
// [32 MT rows] x [128 k] activation slab of its current block in a PRIVATE piece of LDS: two halves of
// 64 k, each row of a half 128 bytes with its 16-byte chunks XOR-swizzled by the row.  The halves are
// filled by LDS-DMA loads (global -> LDS with no register in between; the swizzle is applied to the
// source address, since a DMA instruction writes 1 KiB of LDS in lane order): as soon as the k-steps of
// a half are done, the same half of the wave's NEXT block is requested into it and has the other
// half's k-steps to arrive.  No other wave touches that LDS, so the main loop has no barrier; the
// DMA's completion is awaited with counted s_waitcnt vmcnt (vector-memory operations of a wave complete
// in issue order; the compiler does not track data that arrives in LDS this way).
template <int MT, int T, int KW, int SZS, bool ACT = false>
__global__ void __launch_bounds__(64 * KW) gemm_wide_kernel(int n_slices, int m_blocks, int panel_rows,
                                                            const GemmArgs ga) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub (and has no buffer-resource type)
  fetch_args_now(ga);
  asm volatile("" ::"s"(n_slices), "s"(m_blocks), "s"(panel_rows));
  if (ga.mask != nullptr && ga.mask[blockIdx.z] == 0) return;
  const GemmArgs a = batch_slice(ga);
  constexpr int R = MT * 32;             // rows of the workgroup's tile
  constexpr int HALF = R * 128;          // bytes of a half slab: R rows x 64 k
  constexpr int SLAB = 2 * HALF;         // LDS per wave
  constexpr int XLR = R / 8;             // DMA loads per half slab (8 rows x 128 bytes each)
  constexpr int XL = (W4A16_DBG & 1) ? 1 : XLR;
  constexpr int NSZ = 4 / SZS;           // scale/zero words per 128-k block and tile
  constexpr int WL = T * (NSZ + 2);      // register loads per weight block
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r32 = lane & 31, h = lane >> 5;
#ifdef W4A16_STAMPS
  constexpr int NWAVES = KW;
  int n_done = 0;
  if (lane == 0 && STAMP_OK) g_stamps[((size_t)blockIdx.x * NWAVES + w) * 32 + 30] = __builtin_amdgcn_s_memrealtime();
#endif
  STAMP(0);

  // workgroup id -> (column slice, row block): row blocks of a slice share an XCD (ids 8 apart), in
  // panels once there are more than 8 of them (as gemm_tiled_kernel)
  int slice, mb;
  if (!map_workgroup(n_slices, m_blocks, panel_rows, a.mb_magic, slice, mb)) return;  // whole workgroup
  // A 32-column tile of the matrix instruction is two 16-column tiles of the packed layout: normally neighbours.
  // With the fused activation the weights are [gate | up] (N / 2 columns each, whole 16-column tiles) and tile
  // j of the workgroup is made of gate columns 16 j .. 16 j + 15 and the SAME columns of the up half, so that a
  // lane holds both operands of silu(gate) * up for its columns (accumulator registers 8 apart).
  const int ct0 = slice * T;  // first tile
  const int m0 = mb * R;
  auto tile16 = [&](int t) {
    const int second = (lane >> 4) & 1;
    if constexpr (ACT) {
      const int nth = a.NT / 2;  // 16-column tiles per half
      return min(ct0 + t, nth - 1) + second * nth;
    }
    // 16-column tiles past the end re-read the last one (valid memory); their results are dropped
    return min(2 * (ct0 + t) + second, a.NT - 1);
  };

  // Per-lane byte offsets from the (uniform) buffer bases: 32 bits each (the host checks the sizes).
  uint32_t woff[T], szoff[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int t16 = tile16(t);
    woff[t] = ((uint32_t)t16 * a.KBS * 64 + ((lane & 15) + 16 * h)) * 16u;
    szoff[t] = ((uint32_t)t16 * a.NG * 16 + (lane & 15)) * 4u;
  }
  // DMA p of a half slab fills rows 8 p .. 8 p + 7: lane -> row 8 p + (lane >> 3), 16-byte position
  // lane & 7, which holds chunk (lane & 7) ^ ((row >> 1) & 7) of the row's 64 k
  uint32_t xoff[XLR];
  int xchunk[2];  // chunk index fetched by this lane, p even / odd
#pragma unroll
  for (int e = 0; e < 2; ++e) xchunk[e] = (lane & 7) ^ ((4 * e + (lane >> 4)) & 7);
#pragma unroll
  for (int p = 0; p < XLR; ++p)
    xoff[p] = ((uint32_t)min(m0 + 8 * p + (lane >> 3), a.M - 1) * a.K + xchunk[p & 1] * 8) * 2u;
  const int wbase = w * SLAB;
  // B fragment of k-step ks (0..3) of a half: row r32 (+ 32 mt), chunk 2 ks + h
  int xrd[4];
  {
    const int f = (r32 >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) xrd[ks] = wbase + r32 * 128 + (((2 * ks + h) ^ f) << 4);
  }

  f32x16 acc[T][MT];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][mt][i] = 0.f;
  const DqConst dqc = make_dq_const();

  struct WSet {
    uint32_t sz[T][NSZ];
    u32x4 q[T][2];
  };
  // Buffer-addressed loads throughout: descriptor base + 32-bit lane offset + scalar offset costs no
  // vector arithmetic per load, and (unlike the global_load ... lds form, which the compiler treats as a
  // flat access that forces every later wait to zero) the LDS-DMA variant keeps the compiler's own wait
  // counts for the weight registers exact.
  const __amdgpu_buffer_rsrc_t w_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(a.pw), 0, (int)((long long)a.NT * a.KBS * 1024), 0x00020000);
  const __amdgpu_buffer_rsrc_t sz_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.psz), 0, a.NT * a.NG * 64, 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(a.x), 0, a.M * a.K * 2, 0x00020000);
  auto load_w = [&](int kb, WSet& b) {
    if (W4A16_DBG & 2) kb = w;  // ablation: weights from cache
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int s = 0; s < NSZ; ++s) {
        const int g = min(group_of<SZS>(a, kb * 128 + s * SZS * 32), a.NG - 1);
        b.sz[t][s] = __builtin_amdgcn_raw_buffer_load_b32(sz_rs, szoff[t], g * 64, 0);
      }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      b.q[t][0] = __builtin_amdgcn_raw_buffer_load_b128(w_rs, woff[t], kb * 1024, 0);
      b.q[t][1] = __builtin_amdgcn_raw_buffer_load_b128(w_rs, woff[t] + 512, kb * 1024, 0);
    }
  };
  // XL DMA loads: half hp of block kb's slab -> this wave's LDS half hp (K is a multiple of 128 here)
  // XL DMA loads: half hp of block kb's slab -> this wave's LDS half hp (K is a multiple of 128 here)
  auto dma_x = [&](int kb, int hp) {
#pragma unroll
    for (int p = 0; p < XL; ++p)  // (ablation 1: XL is 1, an eighth of the activation traffic)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          x_rs, (__attribute__((address_space(3))) void*)(smem_raw + wbase + hp * HALF + p * 1024), 16, xoff[p],
          (kb * 128 + hp * 64) * 2, 0, 0);
  };
  // B fragments of k-step `kstep` (0..7) of the block: 2 ds_read_b128 from half kstep / 4
  auto read_x = [&](int kstep, f16x8 (&xf)[MT]) {
    const int hp = kstep >> 2, ks = kstep & 3;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      xf[mt] = *reinterpret_cast<const f16x8*>(smem_raw + xrd[ks] + hp * HALF + mt * 4096);
  };
  // The weight fragment of tile t at k-step `kstep` (scale/zero constants refreshed where a new group starts)
  auto convert = [&](int kstep, int t, const WSet& b, SZ (&c)[T]) {
    if (kstep % (2 * SZS) == 0) c[t] = expand_sz(b.sz[t][kstep / (2 * SZS)]);
    if (W4A16_DBG & 4) return __builtin_bit_cast(f16x8, b.q[t][kstep & 1]);  // ablation: no conversion
    return dequant_word(b.q[t][kstep & 1][kstep >> 1], c[t], dqc);
  };
  // One k-step: MT x T matrix instructions.  The conversion runs ONE TILE AHEAD of them (`af0` comes in
  // converted; tiles 1.. and tile 0 of the next k-step are converted here), and the two instruction kinds
  // are dealt out alternately: a wave issues in order, so a matrix instruction that finds the pipe still
  // busy with its predecessor would otherwise hold back the ~52 cycles of conversion behind it, and a
  // single wave would run at half the matrix rate.
  auto math = [&](int kstep, const WSet& b, SZ (&c)[T], const f16x8 (&xf)[MT], f16x8& af0) {
    f16x8 af[T + 1];
    af[0] = af0;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (t + 1 < T)
        af[t + 1] = convert(kstep, t + 1, b, c);
      else if (kstep + 1 < 8)
        af[T] = convert(kstep + 1, 0, b, c);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[t][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t], xf[mt], acc[t][mt], 0, 0, 0);
    }
    if (kstep + 1 < 8) af0 = af[T];
    if (W4A16_ILV) {
      constexpr int per = (13 + MT - 1) / MT;  // vector instructions of one conversion per matrix instruction
#pragma unroll
      for (int i = 0; i < T * MT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one matrix instruction
        __builtin_amdgcn_sched_group_barrier(0x002, per, 0);  // a share of the next conversion
      }
    }
  };
  // One 128-k block from weight set `cur`.  Unless it is the wave's last (LAST), the next block (kb_next)
  // is requested meanwhile: its weights into `nxt` at the start, each half of its slab into the LDS half
  // just consumed.  A wave's vector-memory operations, in issue order, are thus
  //   W(0), D0(0), D1(0), W(1), D0(1), D1(1), W(2), ..          (D = XL DMA loads, W = WL loads)
  // Half 0 of block i is read from the start of the block: all but the XL + WL youngest must be done
  // (XL in the last block, which requests nothing).  Half 1 is first read during k-step 3 (fragments
  // are fetched one k-step ahead), before D0(i+1) is issued: all but the WL youngest (none in the last
  // block).  A half is overwritten only after the matrix instructions of its last k-step were issued,
  // i.e. after its last read returned.  (Three half slabs per wave, the DMA a whole block ahead, were
  // tried where LDS allows: no faster at 4 waves, and slower wherever the extra LDS costs a resident
  // workgroup.)
  auto block = [&](auto last_, int kb_next, const WSet& cur, WSet& nxt) {
    constexpr bool LAST = decltype(last_)::value != 0;
    SZ c[T];
    f16x8 xa[MT], xb[MT];
    if constexpr (!LAST) {
      load_w(kb_next, nxt);
      __builtin_amdgcn_sched_barrier(0);
    }
    wait_vmcnt<LAST ? XL : XL + WL>();
    read_x(0, xa);
    f16x8 af0 = convert(0, 0, cur, c);
#pragma unroll
    for (int k2 = 0; k2 < 8; k2 += 2) {
      // even k-step on xa while xb arrives, odd k-step on xb while xa arrives
      read_x(k2 + 1, xb);
      __builtin_amdgcn_sched_barrier(0);
      math(k2, cur, c, xa, af0);
      __builtin_amdgcn_sched_barrier(0);
      if (k2 + 1 == 3) wait_vmcnt<LAST ? 0 : WL>();
      if (k2 + 2 < 8) read_x(k2 + 2, xa);
      __builtin_amdgcn_sched_barrier(0);
      math(k2 + 1, cur, c, xb, af0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!LAST) {
        if (k2 + 1 == 3) dma_x(kb_next, 0);
        if (k2 + 1 == 7) dma_x(kb_next, 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#ifdef W4A16_STAMPS
    STAMP(2 + min(n_done, 15));
    ++n_done;
#endif
  };

  // This wave's blocks.  With 8 waves the second wave of each SIMD gets its first requests out several
  // thousand cycles after the first (the CU's memory path takes a workgroup's cold first round at a
  // fraction of its steady rate) and would finish that much later; so the first four waves take a head
  // start of two blocks each: blocks 0-7 go to waves 0-3 (w, w + 4), the rest round-robin over all eight
  // (wave w: 8 + w, 16 + w, ..).  `next` steps through a wave's list.
  // Blocks with a successor go two per trip (the weight sets swap roles, nothing in the loop body is
  // conditional: a branch there makes the compiler's own wait counts collapse to "everything"); an odd
  // one out hands its prefetched set over by register copy, so that the last block always reads `w0`
  // and the accumulators never meet from two paths.
  constexpr bool HEAD = (KW == 8) && W4A16_HEAD;
  auto next = [&](int kb) { return (HEAD && kb < 8) ? kb + 4 : kb + KW; };
  int kb = (HEAD && w >= 4) ? 8 + w : w;
  int nblk = 0;
  for (int b = kb; b < a.KB; b = next(b)) ++nblk;
  if (nblk > 0) {
    WSet w0, w1;
    load_w(kb, w0);
    __builtin_amdgcn_sched_barrier(0);
    dma_x(kb, 0);
    __builtin_amdgcn_sched_barrier(0);  // half 0 before half 1: the counted waits rely on the issue order
    dma_x(kb, 1);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(1);
    int i = 0;
    for (; i + 2 < nblk; i += 2) {
      const int k1 = next(kb), k2 = next(k1);
      block(Flag<0>{}, k1, w0, w1);
      block(Flag<0>{}, k2, w1, w0);
      kb = k2;
    }
    if (i + 1 < nblk) {
      kb = next(kb);
      block(Flag<0>{}, kb, w0, w1);
      w0 = w1;
    }
    block(Flag<1>{}, kb, w0, w1);
  }

  // Cross-wave reduction through LDS (the slabs are free once every wave is past this barrier).  A
  // wave's partial result is T * MT * 4 slices of 1 KiB: 4 accumulator registers x 64 lanes, i.e. for
  // tile (t, mt) and register group q the outputs of token 32 mt + r32, columns 32 t + 8 q + 4 h .. + 3.
  // With 8 waves the upper four first park theirs, the lower four add them in place; then every wave
  // sums the remaining sets for its share of the slices and stores.
  constexpr int NSL = T * MT * 4;
  constexpr int LIVE = KW > 4 ? KW / 2 : KW;
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw);
  auto part = [&](int i) {
    const int t = i / (MT * 4), mt = (i >> 2) % MT, q = i & 3;
    return f32x4{acc[t][mt][4 * q], acc[t][mt][4 * q + 1], acc[t][mt][4 * q + 2], acc[t][mt][4 * q + 3]};
  };
  STAMP(20);
  __syncthreads();
  STAMP(21);
  if constexpr (KW > LIVE) {
    if (w >= LIVE) {
#pragma unroll
      for (int i = 0; i < NSL; ++i) red[((w - LIVE) * NSL + i) * 64 + lane] = part(i);
    }
    __syncthreads();
    if (w < LIVE) {
      // all reads first, then the sums back in place: one dependent round trip, not NSL of them
      f32x4 other[NSL];
#pragma unroll
      for (int i = 0; i < NSL; ++i) other[i] = red[(w * NSL + i) * 64 + lane];
#pragma unroll
      for (int i = 0; i < NSL; ++i) red[(w * NSL + i) * 64 + lane] = other[i] + part(i);
    }
  } else {
#pragma unroll
    for (int i = 0; i < NSL; ++i) red[(w * NSL + i) * 64 + lane] = part(i);
  }
  __syncthreads();
  STAMP(22);
  auto total = [&](int i) {
    f32x4 v = red[i * 64 + lane];
#pragma unroll
    for (int s = 1; s < LIVE; ++s) v += red[(s * NSL + i) * 64 + lane];
    return v;
  };
  for (int io = w; io < (ACT ? NSL / 2 : NSL); io += KW) {
    // fused activation: slices q = 0, 1 of a tile are gate columns, q = 2, 3 the same columns of the up half
    const int i = ACT ? (io >> 1) * 4 + (io & 1) : io;
    const f32x4 v = total(i);
    const int t = i / (MT * 4), mt = (i >> 2) % MT, q = i & 3;
    const int m = m0 + mt * 32 + r32;
    if constexpr (ACT) {
      // values are rounded to fp16 first, then the bias is added in fp16, then the activation: what the
      // separate steps on an fp16 GEMM result give
      const f32x4 vu = total(i + 2);
      const int NH = a.N / 2;
      const int n = (ct0 + t) * 16 + 8 * q + 4 * h;
      if (m < a.M && n < NH) {
        f16x4 hg = f16x4{(f16)v.x, (f16)v.y, (f16)v.z, (f16)v.w};
        f16x4 hu = f16x4{(f16)vu.x, (f16)vu.y, (f16)vu.z, (f16)vu.w};
        if (a.bias != nullptr) {
          hg += *reinterpret_cast<const f16x4*>(a.bias + n);
          hu += *reinterpret_cast<const f16x4*>(a.bias + NH + n);
        }
        *reinterpret_cast<f16x4*>(a.out + (size_t)m * NH + n) = silu_mul(hg, hu);
      }
    } else if (const int n = (ct0 + t) * 32 + 8 * q + 4 * h; m < a.M && n < a.N) {
      f16x4 hv = f16x4{(f16)v.x, (f16)v.y, (f16)v.z, (f16)v.w};
      if (a.bias != nullptr) hv += *reinterpret_cast<const f16x4*>(a.bias + n);
      f16x4* dst = reinterpret_cast<f16x4*>(a.out + (size_t)m * a.N + n);
      if (a.add_out) hv = *dst + hv;  // fp16 + fp16; one lane owns each element
      *dst = hv;
    }
  }
  STAMP(23);
#ifdef W4A16_STAMPS
  if (lane == 0 && STAMP_OK) g_stamps[((size_t)blockIdx.x * NWAVES + w) * 32 + 31] = __builtin_amdgcn_s_memrealtime();
#endif
#endif
}

// ---------------------------------------------------------------------------
// 32x32x16 GEMM for many rows: one activation slab shared by all waves of the workgroup
// ---------------------------------------------------------------------------
// With hundreds of rows and more the workgroup's traffic, not its arithmetic, decides: a [32 MT] x [32 T NW]
// tile re-reads its activation rows once per workgroup, so the wider the tile the fewer bytes per
// matrix instruction.  Here the NW waves of a workgroup each own T 32-column tiles for all MT 32-row
// tiles over the WHOLE of K (no K split, so no cross-wave reduction: results go from the accumulators
// straight to memory), and share the [32 MT rows] x [128 k] activation slab of the current block, which
// the waves fetch together by LDS-DMA into a ring of three slabs (256-byte rows, 16-byte chunks
// XOR-swizzled by the row), two blocks ahead.  A weight fragment converted once feeds MT matrix
// instructions: with MT = 4 the conversion takes a quarter of the vector issue slots of the 16x16x32
// kernels'.  One workgroup barrier per 128-k block: after it every wave's share of the block's slab has
// landed (each waits for its own DMA before arriving) and the slab of the block before last is free.
// Slabs in the ring of gemm_wide_shared_kernel: three (the DMA runs two blocks ahead) for workgroups that have a CU to
// themselves; two for the 4-wave shape that leaves registers and LDS for a second workgroup on the CU, whose
// arithmetic then covers the first one's waits, prologue and stores.
#ifndef W4A16_RING2
#define W4A16_RING2 1
#endif
template <int MT, int T, int NW>
constexpr int ring_slabs() { return (W4A16_RING2 && NW == 4 && MT * T <= 8) ? 2 : 3; }

template <int MT, int T, int NW, int SZS, bool ACT = false>
__global__ void __launch_bounds__(64 * NW, (ring_slabs<MT, T, NW>() == 2 ? 2 : 1)) gemm_wide_shared_kernel(int n_slices, int m_blocks, int panel_rows,
                                                                   const GemmArgs ga) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub
  fetch_args_now(ga);
  asm volatile("" ::"s"(n_slices), "s"(m_blocks), "s"(panel_rows));
  if (ga.mask != nullptr && ga.mask[blockIdx.z] == 0) return;
  const GemmArgs a = batch_slice(ga);
  constexpr int R = MT * 32;
  constexpr int SLAB = R * 256;             // bytes: R rows x 128 k
  constexpr int PIECES = R / 4;             // DMA instructions per slab (4 rows x 256 bytes each)
  static_assert(PIECES % NW == 0, "every wave fetches the same number of pieces");
  constexpr int ND = PIECES / NW;           // ... per wave
  constexpr int NSZ = 4 / SZS;
  constexpr int WL = T * (NSZ + 2);         // register loads per weight block and wave
  constexpr int RING = ring_slabs<MT, T, NW>();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r32 = lane & 31, h = lane >> 5;

  int slice, mb;
  if (!map_workgroup(n_slices, m_blocks, panel_rows, a.mb_magic, slice, mb)) return;  // whole workgroup
  // fused activation: a tile is 16 gate columns and the same 16 columns of the up half (as in gemm_wide_kernel)
  const int ct0 = (slice * NW + w) * T;  // this wave's first tile
  const int m0 = mb * R;
  auto tile16 = [&](int t) {
    const int second = (lane >> 4) & 1;
    if constexpr (ACT) {
      const int nth = a.NT / 2;
      return min(ct0 + t, nth - 1) + second * nth;
    }
    return min(2 * (ct0 + t) + second, a.NT - 1);
  };

  uint32_t woff[T], szoff[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int t16 = tile16(t);
    woff[t] = ((uint32_t)t16 * a.KBS * 64 + ((lane & 15) + 16 * h)) * 16u;
    szoff[t] = ((uint32_t)t16 * a.NG * 16 + (lane & 15)) * 4u;
  }
  // this wave's pieces of a slab: piece p = w + NW j covers rows 4 p .. 4 p + 3; lane -> row 4 p + (lane >> 4),
  // 16-byte position lane & 15, which holds chunk (lane & 15) ^ (row & 15) of the row's 128 k
  uint32_t xoff[ND];
#pragma unroll
  for (int j = 0; j < ND; ++j) {
    const int row = 4 * (w + NW * j) + (lane >> 4);
    xoff[j] = ((uint32_t)min(m0 + row, a.M - 1) * a.K + ((lane & 15) ^ (row & 15)) * 8) * 2u;
  }
  // B fragment of k-step ks (0..7): row r32 (+ 32 mt), chunk 2 ks + h
  const int xrd = r32 * 256;
  const int xsw = r32 & 15;

  f32x16 acc[T][MT];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][mt][i] = 0.f;
  const DqConst dqc = make_dq_const();

  const __amdgpu_buffer_rsrc_t w_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(a.pw), 0, (int)((long long)a.NT * a.KBS * 1024), 0x00020000);
  const __amdgpu_buffer_rsrc_t sz_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.psz), 0, a.NT * a.NG * 64, 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(a.x), 0, a.M * a.K * 2, 0x00020000);
  struct WSet {
    uint32_t sz[T][NSZ];
    u32x4 q[T][2];
  };
  auto load_w = [&](int kb, WSet& b) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int s = 0; s < NSZ; ++s) {
        const int g = min(group_of<SZS>(a, kb * 128 + s * SZS * 32), a.NG - 1);
        b.sz[t][s] = __builtin_amdgcn_raw_buffer_load_b32(sz_rs, szoff[t], g * 64, 0);
      }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      b.q[t][0] = __builtin_amdgcn_raw_buffer_load_b128(w_rs, woff[t], kb * 1024, 0);
      b.q[t][1] = __builtin_amdgcn_raw_buffer_load_b128(w_rs, woff[t] + 512, kb * 1024, 0);
    }
  };
  auto dma_x = [&](int kb, int buf) {
#pragma unroll
    for (int j = 0; j < ND; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          x_rs, (__attribute__((address_space(3))) void*)(smem_raw + buf * SLAB + (w + NW * j) * 1024), 16, xoff[j],
          kb * 256, 0, 0);
  };
  auto read_x = [&](int ks, int base, f16x8 (&xf)[MT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      xf[mt] = *reinterpret_cast<const f16x8*>(smem_raw + base + (((2 * ks + h) ^ xsw) << 4) + mt * 8192);
  };
  auto convert = [&](int kstep, int t, const WSet& b, SZ (&c)[T]) {
    if (kstep % (2 * SZS) == 0) c[t] = expand_sz(b.sz[t][kstep / (2 * SZS)]);
    return dequant_word(b.q[t][kstep & 1][kstep >> 1], c[t], dqc);
  };
  auto math = [&](int kstep, const WSet& b, SZ (&c)[T], const f16x8 (&xf)[MT], f16x8& af0) {
    f16x8 af[T + 1];
    af[0] = af0;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (t + 1 < T)
        af[t + 1] = convert(kstep, t + 1, b, c);
      else if (kstep + 1 < 8)
        af[T] = convert(kstep + 1, 0, b, c);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[t][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t], xf[mt], acc[t][mt], 0, 0, 0);
    }
    if (kstep + 1 < 8) af0 = af[T];
    constexpr int per = (13 + MT - 1) / MT;
#pragma unroll
    for (int i = 0; i < T * MT; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, per, 0);
    }
  };
  // Block i from weight set `cur` in ring slab i % 3.  A wave's vector-memory operations, in issue order:
  //   D(0), D(1), W(0) | D(2), W(1) | D(3), W(2) | ..      (D = its ND DMA loads of a slab, W = WL loads)
  // one group per block, issued right after the block's barrier.  Before the barrier of block i the wave's
  // own share of D(i) must have landed: all but the ND + WL youngest (D(i+1), W(i)) are done then; in the
  // last block D(i+1) does not exist: all but the WL youngest.
  // DMA 0/1: whether D(i+2) exists; LW: whether W(i+1) exists.
  auto block = [&](auto dma, auto lw, int i, const WSet& cur, WSet& nxt) {
    constexpr bool DMA = decltype(dma)::value, LW = decltype(lw)::value;
    const int base = (i % RING) * SLAB + xrd;
    // (two slabs: D(i+1) is issued after this barrier, so only W(i) is younger than D(i))
    wait_vmcnt<(LW && RING == 3) ? ND + WL : WL>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DMA) dma_x(i + RING - 1, (i + RING - 1) % RING);
    __builtin_amdgcn_sched_barrier(0);  // the wait counts above rely on this issue order: DMA first, then weights
    if constexpr (LW) load_w(i + 1, nxt);
    __builtin_amdgcn_sched_barrier(0);
    SZ c[T];
    f16x8 xa[MT], xb[MT];
    read_x(0, base, xa);
    f16x8 af0 = convert(0, 0, cur, c);
#pragma unroll
    for (int k2 = 0; k2 < 8; k2 += 2) {
      read_x(k2 + 1, base, xb);
      __builtin_amdgcn_sched_barrier(0);
      math(k2, cur, c, xa, af0);
      __builtin_amdgcn_sched_barrier(0);
      if (k2 + 2 < 8) read_x(k2 + 2, base, xa);
      __builtin_amdgcn_sched_barrier(0);
      math(k2 + 1, cur, c, xb, af0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  const int n = a.KB;  // >= 2 (host-checked)
  WSet w0, w1;
  dma_x(0, 0);
  if constexpr (RING == 3) dma_x(1, 1);
  __builtin_amdgcn_sched_barrier(0);
  load_w(0, w0);
  __builtin_amdgcn_sched_barrier(0);
  int i = 0;
  for (; i + 2 <= n - 2; i += 2) {
    block(Flag<1>{}, Flag<1>{}, i, w0, w1);
    block(Flag<1>{}, Flag<1>{}, i + 1, w1, w0);
  }
  if (i < n - 2) {
    block(Flag<1>{}, Flag<1>{}, i, w0, w1);
    w0 = w1;
    ++i;
  }
  block(Flag<RING == 2>{}, Flag<1>{}, i, w0, w1);  // (two slabs: the last block's slab is still to be fetched)
  block(Flag<0>{}, Flag<0>{}, i + 1, w1, w0);

  // straight from the accumulators: register group q of tile (t, mt) is token 32 mt + r32, columns 8 q + 4 h .. + 3
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m0 + mt * 32 + r32;
        if constexpr (ACT) {
          // register groups 0, 1 are gate columns, 2, 3 the same columns of the up half; rounded to fp16,
          // bias added in fp16, then the activation, as the separate steps would
          const int NH = a.N / 2;
          const int nn = (ct0 + t) * 16 + 8 * q + 4 * h;
          if (q < 2 && m < a.M && nn < NH) {
            constexpr int up = ACT ? 8 : 0;
            f16x4 hg = f16x4{(f16)acc[t][mt][4 * q], (f16)acc[t][mt][4 * q + 1], (f16)acc[t][mt][4 * q + 2],
                             (f16)acc[t][mt][4 * q + 3]};
            f16x4 hu = f16x4{(f16)acc[t][mt][(4 * q + up) & 15], (f16)acc[t][mt][(4 * q + 1 + up) & 15],
                             (f16)acc[t][mt][(4 * q + 2 + up) & 15], (f16)acc[t][mt][(4 * q + 3 + up) & 15]};
            if (a.bias != nullptr) {
              hg += *reinterpret_cast<const f16x4*>(a.bias + nn);
              hu += *reinterpret_cast<const f16x4*>(a.bias + NH + nn);
            }
            *reinterpret_cast<f16x4*>(a.out + (size_t)m * NH + nn) = silu_mul(hg, hu);
          }
        } else if (const int nn = (ct0 + t) * 32 + 8 * q + 4 * h; m < a.M && nn < a.N) {
          f16x4 hv = f16x4{(f16)acc[t][mt][4 * q], (f16)acc[t][mt][4 * q + 1], (f16)acc[t][mt][4 * q + 2],
                           (f16)acc[t][mt][4 * q + 3]};
          if (a.bias != nullptr) hv += *reinterpret_cast<const f16x4*>(a.bias + nn);
          f16x4* dst = reinterpret_cast<f16x4*>(a.out + (size_t)m * a.N + nn);
          if (a.add_out) hv = *dst + hv;  // fp16 + fp16; one lane owns each element
          *dst = hv;
        }
      }
#endif
}

// ---------------------------------------------------------------------------
// Shape-agnostic GEMM on the stock layout (no repack, no workspace)
// ---------------------------------------------------------------------------
// The stock layout keeps the 8 nibbles of a word on 8 different output columns of ONE k; the matrix
// instruction wants 8 consecutive k of one column in a lane.  A lane therefore loads the same packed word
// column from 8 rows and transposes the 8 x 8 nibbles in registers (three rounds of masked swaps, about 50
// instructions per 64 weights): word j of the result holds 8 k of the column in nibble j, in the nibble
// order the private format uses, so the conversion and its constants are the ones of the packed kernels and
// the weights fed to the matrix cores are bit-identical to theirs.
//
// A wavefront owns 16 adjacent word columns (lane & 15) x the four k-octets of a 32-k step (lane >> 4), i.e.
// 128 output columns as 8 matrix tiles whose 16 "rows" are 8 columns apart; tile j's accumulator register v
// of lane (token, q) is output column 8 (16 bx + 4 q + v) + (column of nibble j), so a lane ends up with 8
// adjacent columns per register index and stores 16 bytes.  The KW wavefronts of a workgroup split K and
// add their accumulators pairwise through LDS (log2 KW rounds).
template <int MT, int KW>
__global__ void __launch_bounds__(KW * 64) gemm_stock_kernel(const f16* __restrict__ x,
                                                             const uint32_t* __restrict__ qw,
                                                             const f16* __restrict__ scales,
                                                             const uint32_t* __restrict__ qz,
                                                             f16* __restrict__ out, int M, int K,
                                                             int N, int G) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i16 = lane & 15, q = lane >> 4;
  const int NC = N / 8;
  const int c = min((int)blockIdx.x * 16 + i16, NC - 1);  // past the edge: a valid column, never stored
  const int m0 = blockIdx.y * (16 * MT);
  const DqConst dqc = make_dq_const();

  const f16* xp[MT];
  bool xok[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + 16 * mt + i16;
    xok[mt] = m < M;
    xp[mt] = x + (size_t)(xok[mt] ? m : 0) * K + 8 * q;
  }
  f32x4 acc[8][MT];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[j][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // nibble j of a stock word is logical column kCol[j] of its 8; row kCol[p] of the octet goes to nibble p
  constexpr int kCol[8] = {0, 2, 4, 6, 1, 3, 5, 7};
  const int nsteps = K / 32;
  const int per = (nsteps + KW - 1) / KW;
  const int s_end = min(nsteps, (w + 1) * per);
  int g_cur = -1;
  SZ sz[8];
  for (int s = w * per; s < s_end; ++s) {
    const int k0 = s * 32 + 8 * q;
    uint32_t r[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) r[p] = __builtin_nontemporal_load(qw + (size_t)(k0 + kCol[p]) * NC + c);
    f16x8 xf[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      xf[mt] = *reinterpret_cast<const f16x8*>(xp[mt] + s * 32);
      if (!xok[mt]) xf[mt] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    const int g = (s * 32) / G;
    if (g != g_cur) {  // uniform over the wavefront
      g_cur = g;
      const uint32_t zw = qz[(size_t)g * NC + c];
      const u32x4 sc = *reinterpret_cast<const u32x4*>(scales + (size_t)g * N + 8 * c);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t z = (zw >> (4 * j)) & 0xfu;
        const uint32_t sbits = (sc[kCol[j] >> 1] >> (16 * (kCol[j] & 1))) & 0xffffu;
        sz[j] = expand_sz(sbits | ((0xE400u | z) << 16));  // fp16 -(1024 + z), as the packer writes it
      }
    }
    // 8 x 8 nibble transpose: swap the off-diagonal blocks of size 4, 2, 1
    auto swap_blocks = [&](auto b_, uint32_t mask) {
      constexpr int b = decltype(b_)::value;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i & b) continue;
        const uint32_t t = ((r[i] >> (4 * b)) ^ r[i + b]) & mask;
        r[i + b] ^= t;
        r[i] ^= t << (4 * b);
      }
    };
    swap_blocks(Flag<4>{}, 0x0000ffffu);
    swap_blocks(Flag<2>{}, 0x00ff00ffu);
    swap_blocks(Flag<1>{}, 0x0f0f0f0fu);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f16x8 af = dequant_word(r[j], sz[j], dqc);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, xf[mt], acc[j][mt], 0, 0, 0);
    }
  }

  // pairwise sums over the K split: the upper half of the live wavefronts parks, the lower half adds
  f32x4* park = reinterpret_cast<f32x4*>(smem);
#pragma unroll
  for (int half = KW / 2; half >= 1; half >>= 1) {
    if (w >= half && w < 2 * half) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) park[((w - half) * 8 * MT + j * MT + mt) * 64 + lane] = acc[j][mt];
    }
    __syncthreads();
    if (w < half) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[j][mt] += park[(w * 8 * MT + j * MT + mt) * 64 + lane];
    }
    __syncthreads();
  }
  if (w != 0) return;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + 16 * mt + i16;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int cw = blockIdx.x * 16 + 4 * q + v;
      if (m < M && cw < NC) {
        f16x8 h;
#pragma unroll
        for (int j = 0; j < 8; ++j) h[kCol[j]] = (f16)acc[j][mt][v];
        *reinterpret_cast<f16x8*>(out + (size_t)m * N + 8 * cw) = h;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Dequantize (stock layout) : one thread per packed word -> 16-byte store
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) dequant_stock_kernel(const uint32_t* __restrict__ qw,
                                                            const f16* __restrict__ scales,
                                                            const uint32_t* __restrict__ qz,
                                                            f16* __restrict__ out, int K, int N,
                                                            int G) {
  const int NC = N / 8;
  const size_t total = (size_t)K * NC;
  const f16x2 sixteenth = f16x2{(f16)0.0625f, (f16)0.0625f};
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * 256) {
    const int k = (int)(idx / NC);
    const int c = (int)(idx - (size_t)k * NC);
    const int g = k / G;
    const uint32_t q = qw[idx];
    const uint32_t zw = qz[(size_t)g * NC + c];
    const u32x4 sw = *reinterpret_cast<const u32x4*>(scales + (size_t)g * N + 8 * c);
    const uint32_t t = q >> 8, zt = zw >> 8;
    const f16x2 d0 = (as_h2((q & kLoMask) | kMagic) - as_h2((zw & kLoMask) | kMagic)) * as_h2(sw.x);
    const f16x2 d1 = ((as_h2((q & kHiMask) | kMagic) - as_h2((zw & kHiMask) | kMagic)) * sixteenth) *
                     as_h2(sw.y);
    const f16x2 d2 = (as_h2((t & kLoMask) | kMagic) - as_h2((zt & kLoMask) | kMagic)) * as_h2(sw.z);
    const f16x2 d3 = ((as_h2((t & kHiMask) | kMagic) - as_h2((zt & kHiMask) | kMagic)) * sixteenth) *
                     as_h2(sw.w);
    const u32x4 o = u32x4{as_u32(d0), as_u32(d1), as_u32(d2), as_u32(d3)};
    __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(out + (size_t)k * N + 8 * c));
  }
}

// ---------------------------------------------------------------------------
// Re-layout: stock -> private
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) repack_weight_kernel(const uint32_t* __restrict__ qw,
                                                            const uint32_t* __restrict__ qz,
                                                            uint32_t* __restrict__ pw, int K, int N,
                                                            int G, int NT, int KB) {  // KB: padded count
  const int NC = N / 8;
  const size_t total = (size_t)NT * KB * 256;
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total;
       o += (size_t)gridDim.x * 256) {
    const int j = (int)(o & 3);
    const int l = (int)((o >> 2) & 63);
    const size_t blk = o >> 8;
    const int kb = (int)(blk % KB);
    const int nt = (int)(blk / KB);
    const int n = nt * 16 + (l & 15);
    const int k0 = kb * 128 + j * 32 + (l >> 4) * 8;
    uint32_t word = 0;
    if (n < N) {
      const int src_shift = 4 * kOrderDev[n & 7];
      // k >= K (K not a multiple of 128) is filled with the last group's zero point, so
      // those weights dequantise to exactly 0 whatever activation they meet.
      const uint32_t zpad = (qz[(size_t)(K / G - 1) * NC + (n >> 3)] >> src_shift) & 0xfu;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = k0 + i;
        const uint32_t v =
            (k < K) ? ((qw[(size_t)k * NC + (n >> 3)] >> src_shift) & 0xfu) : zpad;
        word |= v << (4 * kOrderDev[i]);
      }
    }
    pw[o] = word;
  }
}

__global__ void __launch_bounds__(256) repack_sz_kernel(const f16* __restrict__ scales,
                                                        const uint32_t* __restrict__ qz,
                                                        uint32_t* __restrict__ psz, int N, int NT,
                                                        int NG) {
  const int NC = N / 8;
  const size_t total = (size_t)NT * NG * 16;
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total;
       o += (size_t)gridDim.x * 256) {
    const int n16 = (int)(o & 15);
    const size_t r = o >> 4;
    const int g = (int)(r % NG);
    const int nt = (int)(r / NG);
    const int n = nt * 16 + n16;
    uint32_t word = 0;
    if (n < N) {
      const uint32_t z = (qz[(size_t)g * NC + (n >> 3)] >> (4 * kOrderDev[n & 7])) & 0xfu;
      const uint32_t sbits = *reinterpret_cast<const uint16_t*>(scales + (size_t)g * N + n);
      word = sbits | ((0xE400u | z) << 16);  // fp16 -(1024 + z)
    }
    psz[o] = word;
  }
}

// Dequantize from the private layout (needed when the dense large-batch route is taken after the
// stock tensors were dropped).  A wave takes one 128-k block of 8 adjacent column tiles: thread
// (tile, half, octet) loads the 16-byte lane quads of 8 adjacent columns (all four k-steps of one
// k-octet), converts them and transposes 8 columns x 8 k in registers, so that every store is
// 16 bytes of one output row and a store instruction covers 4 rows x 256 contiguous bytes.
__global__ void __launch_bounds__(256) dequant_packed_kernel(const uint32_t* __restrict__ pw,
                                                             const uint32_t* __restrict__ psz,
                                                             f16* __restrict__ out, int K, int N,
                                                             int G, int NT, int KB, int NG) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int ntg = blockIdx.x;                 // group of 8 column tiles
  const int o = lane >> 4;                    // k-octet of a 32-k step: 16 adjacent lanes share a row
  const int h = lane & 1;                     // which 8 of the tile's 16 columns
  const int nt = ntg * 8 + ((lane >> 1) & 7);
  const int n0 = nt * 16 + h * 8;
  if (nt >= NT || n0 >= N) return;
  const u32x4* blocks = reinterpret_cast<const u32x4*>(pw) + (size_t)nt * KB * 64 + (o * 16 + h * 8);
  const uint32_t* sz = psz + (size_t)nt * NG * 16 + h * 8;
  for (int kb = blockIdx.y * 4 + wave; kb < KB; kb += gridDim.y * 4) {
    const int kbase = kb * 128 + o * 8;
    if (kb * 128 >= K) break;
    u32x4 q[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) q[c] = blocks[(size_t)kb * 64 + c];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k0 = kbase + j * 32;
      if (k0 >= K) break;
      const int g = k0 / G;
      f16x8 v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = dequant_word(q[c][j], expand_sz(sz[g * 16 + c]));
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f16x8 row = f16x8{v[0][i], v[1][i], v[2][i], v[3][i], v[4][i], v[5][i], v[6][i], v[7][i]};
        __builtin_nontemporal_store(row, reinterpret_cast<f16x8*>(out + (size_t)(k0 + i) * N + n0));
      }
    }
  }
}

// Measurement aid: reads `n16` 16-byte words once, as the GEMM kernels read their weights (one 1-KiB
// wave-instruction after another, non-temporal), and keeps one xor of them per workgroup so that the loads
// cannot be dropped.  Its duration on a buffer of a layer's size is the floor for that layer's decode launch.
__global__ void __launch_bounds__(256) stream_read_kernel(const u32x4* __restrict__ src, size_t n16,
                                                          uint32_t* __restrict__ sink) {
  u32x4 acc = u32x4{0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
    const u32x4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
    acc ^= a ^ b ^ c ^ d;
  }
  for (; i < n16; i += stride) acc ^= __builtin_nontemporal_load(src + i);
  const uint32_t v = acc.x ^ acc.y ^ acc.z ^ acc.w;
  if (v == 0x9e3779b9u) sink[blockIdx.x] = v;  // practically never: the data decides, so the loads stay
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
thread_local int g_last_hip_error = 0;

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// The packed layout stores a multiple of 4 blocks of 128 k per column tile; the padding holds
// zero-valued weights (q = z), so kernels whose K-groups take blocks in lockstep never branch.
inline int padded_kblocks(int K) { return ceil_div(K, 512) * 4; }

int check_common(int K, int N, int G) {
  if (K <= 0 || N <= 0 || G <= 0) return W4A16_E_SHAPE;
  if (N % 8) return W4A16_E_OC_PACK;
  if (G % 32) return W4A16_E_GROUP32;
  if (K % G) return W4A16_E_K_GROUP;
  if ((long long)K * (N / 8) >= (1LL << 31)) return W4A16_E_TOO_LARGE;
  return W4A16_OK;
}

inline bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }

// hipGetLastError() is sticky per thread: drop whatever an earlier, unrelated call left
// behind so that launched() reports this library's launches only.
inline void clear_stale_error() { (void)hipGetLastError(); }

int launched() {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    return W4A16_E_LAUNCH;
  }
  return W4A16_OK;
}

// Validates the arguments of the packed-layout GEMM entry points and fills the kernel argument block.
int fill_args(GemmArgs* a, const void* x, const void* packed_w, const void* packed_sz, const void* bias,
              void* out, int M, int K, int N, int G, int batch = 1, long long x_batch_stride = 0,
              long long out_batch_stride = 0) {
  if (!x || !packed_w || !packed_sz || !out) return W4A16_E_NULL;
  if (M <= 0) return W4A16_E_SHAPE;
  const int rc = check_common(K, N, G);
  if (rc) return rc;
  if (K % 32) return W4A16_E_SHAPE;
  if ((long long)M * N >= (1LL << 31) || (long long)M * K >= (1LL << 31)) return W4A16_E_TOO_LARGE;
  if (misaligned(x) || misaligned(packed_w) || misaligned(packed_sz)) return W4A16_E_ALIGN;
  if (reinterpret_cast<uintptr_t>(out) & 7u) return W4A16_E_ALIGN;
  if (bias && (reinterpret_cast<uintptr_t>(bias) & 7u)) return W4A16_E_ALIGN;
  a->x = (const f16*)x;
  a->pw = (const u32x4*)packed_w;
  a->psz = (const uint32_t*)packed_sz;
  a->bias = (const f16*)bias;
  a->out = (f16*)out;
  a->M = M;
  a->K = K;
  a->N = N;
  a->G = G;
  a->NT = ceil_div(N, 16);
  a->KB = ceil_div(K, 128);
  a->KBS = padded_kblocks(K);
  a->NG = K / G;
  a->add_out = 0;
  a->mask = nullptr;
  a->x_bs = a->out_bs = a->pw_bs = a->psz_bs = a->bias_bs = 0;
  a->batched = batch != 1;
  // exact for every block index below 1024 and G / 128 <= 64: (kb * ceil(65536 / d)) >> 16 == kb / d
  a->g_magic = (G % 128 == 0 && G / 128 <= 64 && a->KBS < 1024) ? (65536 + G / 128 - 1) / (G / 128) : 0;
  a->mb_magic = 0;
  if (batch != 1) {
    if (batch < 1 || batch > 65535) return W4A16_E_SHAPE;
    // every problem's buffers must keep the alignment the kernels rely on
    if (x_batch_stride < 0 || x_batch_stride % 8 || out_batch_stride < 0 || out_batch_stride % 4)
      return W4A16_E_SHAPE;
    a->x_bs = x_batch_stride;
    a->out_bs = out_batch_stride;
    a->pw_bs = (long long)a->NT * a->KBS * 64;   // u32x4 units: w4a16_packed_weight_bytes / 16
    a->psz_bs = (long long)a->NT * a->NG * 16;   // uint32 units: w4a16_packed_sz_bytes / 4
    a->bias_bs = bias ? N : 0;
  }
  return W4A16_OK;
}

// The fused-activation variants are built for the row counts the streaming kernel serves by default.
// j / m_blocks by multiply and shift for every work item j of a one-panel launch (exact while j * m_blocks < 65536)
int row_block_magic(long long items, int m_blocks, int panel_rows) {
  if (panel_rows != m_blocks || items * m_blocks >= 65536) return 0;
  return (65536 + m_blocks - 1) / m_blocks;
}

template <int MT, int T>
constexpr bool kActStreaming = (T % 2 == 0) && MT <= 2;

template <int MT, int T, int W>
int launch_packed(const GemmArgs& a, hipStream_t s, int batch, bool act) {
  const size_t lds = (size_t)reduce_lds_tiles(W, T * MT) * 1024;
  const int tiles = act ? a.NT / 2 : a.NT, per_wg = act ? T / 2 : T;
  const dim3 grid(ceil_div(tiles, per_wg), ceil_div(a.M, MT * 16), batch);
  const dim3 block(64 * W);
  void (*kern)(const GemmArgs) = nullptr;
  if (act) {
    if constexpr (kActStreaming<MT, T>) {
      if (a.G % 128 == 0)
        kern = gemm_packed_kernel<MT, T, W, 4, true>;
      else if (a.G == 64)
        kern = gemm_packed_kernel<MT, T, W, 2, true>;
      else
        kern = gemm_packed_kernel<MT, T, W, 1, true>;
    } else {
      return W4A16_E_SHAPE;
    }
  } else if (a.G % 128 == 0)
    kern = gemm_packed_kernel<MT, T, W, 4>;
  else if (a.G == 64)
    kern = gemm_packed_kernel<MT, T, W, 2>;
  else
    kern = gemm_packed_kernel<MT, T, W, 1>;
  if (lds > 64 * 1024) {
    // Per device and per kernel; cheap next to a launch, and legal during stream capture.
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      g_last_hip_error = (int)e;
      return W4A16_E_LAUNCH;
    }
  }
  hipLaunchKernelGGL(kern, grid, block, lds, s, a);
  return launched();
}

template <int MT, int T, int NW, int KW>
constexpr bool kActTiled = (NW % 2 == 0) && (MT == 4 || (MT == 2 && NW == 2)) && T == 3;

template <int MT, int T, int NW, int KW>
int launch_tiled(const GemmArgs& a, hipStream_t s, int batch, bool act) {
  constexpr int stage = 2 * KW * MT * 16 * 256;         // double-buffered slabs
  constexpr int reduce = NW * KW * T * MT * 1024;       // partial tiles
  constexpr int lds = stage > reduce ? stage : reduce;
  static_assert(lds <= 160 * 1024, "LDS budget");
  if (a.KBS % KW) return W4A16_E_SHAPE;
  const int n_slices = act ? ceil_div(a.NT / 2, NW / 2 * T) : ceil_div(a.NT, NW * T);
  const int m_blocks = ceil_div(a.M, MT * 16);
  const int blocks = ceil_div((long long)n_slices * m_blocks, 8) * 8;
  void (*kern)(int, int, int, const GemmArgs) = nullptr;
  if (act) {
    if constexpr (kActTiled<MT, T, NW, KW>) {
      if (a.G % 128 == 0)
        kern = gemm_tiled_kernel<MT, T, NW, KW, 4, true>;
      else if (a.G == 64)
        kern = gemm_tiled_kernel<MT, T, NW, KW, 2, true>;
      else
        kern = gemm_tiled_kernel<MT, T, NW, KW, 1, true>;
    } else {
      return W4A16_E_SHAPE;
    }
  } else if (a.G % 128 == 0)
    kern = gemm_tiled_kernel<MT, T, NW, KW, 4>;
  else if (a.G == 64)
    kern = gemm_tiled_kernel<MT, T, NW, KW, 2>;
  else
    kern = gemm_tiled_kernel<MT, T, NW, KW, 1>;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      g_last_hip_error = (int)e;
      return W4A16_E_LAUNCH;
    }
  }
  // panels of W4A16_PANEL row-blocks once there are more than 8 of them (see the kernel's id mapping)
  const int panel_rows = m_blocks > 8 ? W4A16_PANEL : m_blocks;
  GemmArgs am = a;
  am.mb_magic = row_block_magic((long long)n_slices * m_blocks, m_blocks, panel_rows);
  hipLaunchKernelGGL(kern, dim3(blocks, 1, batch), dim3(64 * NW * KW), lds, s, n_slices, m_blocks, panel_rows, am);
  return launched();
}

// (mt, t, nw, kw) of the LDS-staged kernel
// (the shapes choose_cfg can return; the kernel serves what the 32x32x16 kernels do not take: K not a
// multiple of 128, 24-32 rows of short-K or batched problems)
#define W4A16_FOR_EACH_TILED(X) \
  X(4, 3, 2, 4) X(4, 3, 4, 2) X(4, 3, 1, 4) X(4, 2, 2, 4) X(2, 3, 2, 4) X(2, 3, 1, 4) X(2, 3, 1, 8)

int dispatch_tiled(const GemmArgs& a, int mt, int t, int nw, int kw, hipStream_t s, int batch = 1,
                   bool act = false) {
#define X(MT_, T_, NW_, KW_) \
  if (mt == MT_ && t == T_ && nw == NW_ && kw == KW_) return launch_tiled<MT_, T_, NW_, KW_>(a, s, batch, act);
  W4A16_FOR_EACH_TILED(X)
#undef X
  return W4A16_E_SHAPE;
}

template <int MT, int T, int KW>
int launch_wide(const GemmArgs& a, hipStream_t s, int batch, bool act) {
  constexpr int stage = KW * MT * 32 * 256;  // one private slab per wave
  constexpr int reduce = (KW > 4 ? KW / 2 : KW) * T * MT * 4 * 1024;     // partial tiles
  constexpr int lds = stage > reduce ? stage : reduce;
  static_assert(lds <= 160 * 1024, "LDS budget");
  // whole 128-k blocks only, and 32-bit byte offsets into the three buffers
  if (a.K % 128 || (long long)a.NT * a.KBS * 1024 >= (1LL << 32) || (long long)a.M * a.K >= (1LL << 31)) return W4A16_E_SHAPE;
  // fused activation: a tile is 16 gate + 16 up columns, so there are N / 32 of them too
  if (act && a.N % 32) return W4A16_E_SHAPE;
  const int n_slices = ceil_div(ceil_div(a.N, 32), T);
  const int m_blocks = ceil_div(a.M, MT * 32);
  const int blocks = ceil_div((long long)n_slices * m_blocks, 8) * 8;
  void (*kern)(int, int, int, const GemmArgs) = nullptr;
  if (act) {
    if (a.G % 128 == 0)
      kern = gemm_wide_kernel<MT, T, KW, 4, true>;
    else if (a.G == 64)
      kern = gemm_wide_kernel<MT, T, KW, 2, true>;
    else
      kern = gemm_wide_kernel<MT, T, KW, 1, true>;
  } else if (a.G % 128 == 0)
    kern = gemm_wide_kernel<MT, T, KW, 4>;
  else if (a.G == 64)
    kern = gemm_wide_kernel<MT, T, KW, 2>;
  else
    kern = gemm_wide_kernel<MT, T, KW, 1>;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      g_last_hip_error = (int)e;
      return W4A16_E_LAUNCH;
    }
  }
  const int panel_rows = m_blocks > 8 ? W4A16_PANEL : m_blocks;
  GemmArgs am = a;
  am.mb_magic = row_block_magic((long long)n_slices * m_blocks, m_blocks, panel_rows);
  hipLaunchKernelGGL(kern, dim3(blocks, 1, batch), dim3(64 * KW), lds, s, n_slices, m_blocks, panel_rows, am);
  return launched();
}

// (mt, t, kw) of the 32x32x16 kernel: mt, t in 32-row / 32-column tiles
#define W4A16_FOR_EACH_WIDE(X) \
  X(2, 3, 8) X(2, 2, 8) X(2, 1, 8) X(1, 3, 8) X(1, 2, 8) X(2, 3, 4) X(2, 2, 4) X(1, 3, 4)

int dispatch_wide(const GemmArgs& a, int mt, int t, int kw, hipStream_t s, int batch = 1, bool act = false) {
#define X(MT_, T_, KW_) \
  if (mt == MT_ && t == T_ && kw == KW_) return launch_wide<MT_, T_, KW_>(a, s, batch, act);
  W4A16_FOR_EACH_WIDE(X)
#undef X
  return W4A16_E_SHAPE;
}

template <int MT, int T, int NW>
int launch_wide_shared(const GemmArgs& a, hipStream_t s, int batch, bool act) {
  constexpr int lds = ring_slabs<MT, T, NW>() * MT * 32 * 256;  // ring of slabs
  static_assert(lds <= 160 * 1024, "LDS budget");
  // whole 128-k blocks, at least two of them, and 32-bit byte offsets into the three buffers
  if (a.K % 128 || a.K < 256 || (long long)a.NT * a.KBS * 1024 >= (1LL << 32) || (long long)a.M * a.K >= (1LL << 31))
    return W4A16_E_SHAPE;
  if (act && a.N % 32) return W4A16_E_SHAPE;
  const int n_slices = ceil_div(ceil_div(a.N, 32), T * NW);
  const int m_blocks = ceil_div(a.M, MT * 32);
  const int blocks = ceil_div((long long)n_slices * m_blocks, 8) * 8;
  void (*kern)(int, int, int, const GemmArgs) = nullptr;
  if (act) {
    if (a.G % 128 == 0)
      kern = gemm_wide_shared_kernel<MT, T, NW, 4, true>;
    else if (a.G == 64)
      kern = gemm_wide_shared_kernel<MT, T, NW, 2, true>;
    else
      kern = gemm_wide_shared_kernel<MT, T, NW, 1, true>;
  } else if (a.G % 128 == 0)
    kern = gemm_wide_shared_kernel<MT, T, NW, 4>;
  else if (a.G == 64)
    kern = gemm_wide_shared_kernel<MT, T, NW, 2>;
  else
    kern = gemm_wide_shared_kernel<MT, T, NW, 1>;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      g_last_hip_error = (int)e;
      return W4A16_E_LAUNCH;
    }
  }
  const int panel_rows = m_blocks > 8 ? W4A16_PANEL : m_blocks;
  GemmArgs am = a;
  am.mb_magic = row_block_magic((long long)n_slices * m_blocks, m_blocks, panel_rows);
  hipLaunchKernelGGL(kern, dim3(blocks, 1, batch), dim3(64 * NW), lds, s, n_slices, m_blocks, panel_rows, am);
  return launched();
}

// (mt, t, nw) of the shared-slab 32x32x16 kernel
#define W4A16_FOR_EACH_WIDE_SHARED(X) X(4, 1, 8) X(4, 2, 8) X(2, 1, 8) X(2, 2, 8) X(2, 3, 8) X(4, 3, 4) X(4, 2, 4)

int dispatch_wide_shared(const GemmArgs& a, int mt, int t, int nw, hipStream_t s, int batch = 1, bool act = false) {
#define X(MT_, T_, NW_) \
  if (mt == MT_ && t == T_ && nw == NW_) return launch_wide_shared<MT_, T_, NW_>(a, s, batch, act);
  W4A16_FOR_EACH_WIDE_SHARED(X)
#undef X
  return W4A16_E_SHAPE;
}

// (mt, t, waves) -> instantiation.  One or two 16-row tiles (more rows go to the other kernels); t in 1..8;
// 16 waves where 128 VGPRs per lane are enough, 4 / 8 for short K and for small workgroups, several to a CU.
#define W4A16_FOR_EACH_CFG(X) \
  X(1, 1, 16) X(1, 2, 16) X(1, 3, 16)                                     \
  X(1, 1, 8)  X(1, 2, 8)  X(1, 3, 8)  X(1, 4, 8)  X(1, 6, 8)  X(1, 8, 8)  \
  X(1, 1, 4)  X(1, 2, 4)  X(1, 3, 4)  X(1, 4, 4)                          \
  X(2, 1, 16)                                                             \
  X(2, 1, 8)  X(2, 2, 8)  X(2, 3, 8)  X(2, 4, 8)  X(2, 6, 8)              \
  X(2, 1, 4)  X(2, 2, 4)  X(2, 3, 4)

int dispatch_packed(const GemmArgs& a, int mt, int t, int waves, hipStream_t s, int batch = 1,
                    bool act = false) {
#define X(MT_, T_, W_) \
  if (mt == MT_ && t == T_ && waves == W_) return launch_packed<MT_, T_, W_>(a, s, batch, act);
  W4A16_FOR_EACH_CFG(X)
#undef X
  return W4A16_E_SHAPE;
}

// ---- decode kernel (up to 4 rows) ----
constexpr int kDecodeMaxRows = 4;    // with LDS staging, the compact reduction and the optional K split
constexpr int kDecodeMaxRowsL2 = 16;  // with fragments straight from L2 (no K split)
constexpr int kMaxSplits = 8;

// What the decode kernel's addressing admits: whole 128-k blocks, 32-bit byte offsets, and a group size whose
// group index is a shift or one multiply away from the block index.
bool decode_ok(const GemmArgs& a) {
  if (a.M > kDecodeMaxRowsL2 || a.K % 128 || (long long)a.M * a.K * 2 >= (1LL << 31)) return false;
  if (!(a.G == 32 || a.G == 64 || (a.G % 128 == 0 && a.G / 128 <= 64))) return false;
  if (a.KB >= 1024) return false;
  if ((long long)a.NT * a.KBS * 1024 >= (1LL << 32) || (long long)a.NT * a.NG * 64 >= (1LL << 31)) return false;
  return true;
}

// Bytes of workspace a K-split launch needs: one slot of fp32 partial sums per (problem, workgroup column,
// split), then one ticket counter per (problem, workgroup column).
size_t decode_partial_bytes(int batch, int gx, int splits, bool act) {
  return (size_t)batch * gx * splits * (act ? 2048 : 1024);
}
// The ticket counters come first, in a region of fixed size: wherever a launch's partial sums land, they never
// overwrite a counter that another launch shape expects to find at zero.
constexpr int kMaxTickets = 16384;
constexpr size_t kTicketBytes = (size_t)kMaxTickets * 4;
size_t decode_ws_bytes(int batch, int gx, int splits, bool act) {
  return splits <= 1 ? 0 : kTicketBytes + decode_partial_bytes(batch, gx, splits, act);
}

// scale/zero words per 128-k block and tile, and the LDS a workgroup needs: per wave 1 KiB of activations per
// block, then the reduction's partial sums
int decode_nsz(int G) { return G % 128 == 0 ? 1 : G == 64 ? 2 : 4; }
size_t decode_lds_bytes(int t, int waves, int nb_max, int M, int G) {
  (void)G;
  if (M > kDecodeMaxRows) return (size_t)waves * t * 1024;  // whole partial tiles, no staged activations
  return (size_t)waves * nb_max * 1024 + (size_t)waves * t * 256;
}

template <int T, int W, int DB>
int launch_decode(const GemmArgs& a, hipStream_t s, int batch, bool act, int splits, void* ws, size_t ws_bytes) {
  constexpr bool kPlain = T <= 8;
  constexpr bool kAct = T % 2 == 0 && T / 2 <= 8;
  if (!decode_ok(a) || (act && a.N % 32)) return W4A16_E_SHAPE;
  if (act ? !kAct : !kPlain) return W4A16_E_SHAPE;  // (before any division by tiles per workgroup: T / 2 is 0 for T = 1)
  if (splits < 1 || splits > kMaxSplits) return W4A16_E_SHAPE;
  const bool xlds = a.M <= kDecodeMaxRows;
  if (!xlds && splits > 1) return W4A16_E_SHAPE;  // the K split is built for up to 4 rows
  if (splits > 1 && (act ? T / 2 : T) > 4) return W4A16_E_SHAPE;  // and up to 4 tiles or tile pairs
  const int kb_per_split = ceil_div(a.KB, splits);
  splits = ceil_div(a.KB, kb_per_split);  // no empty split
  const int nb_max = ceil_div(kb_per_split, W);
  const size_t lds = decode_lds_bytes(T, W, nb_max + 1, a.M, a.G);  // (+ 1: a wave may take a block from another)
  if (lds > 160 * 1024) return W4A16_E_SHAPE;
  const int tiles = act ? a.NT / 2 : a.NT, per_wg = act ? T / 2 : T;
  const int gx = ceil_div(tiles, per_wg);
  f32x4* partial = nullptr;
  unsigned* tickets = nullptr;
  if (splits > 1) {
    if (!ws) return W4A16_E_NULL;
    if (misaligned(ws)) return W4A16_E_ALIGN;
    if (ws_bytes < decode_ws_bytes(batch, gx, splits, act)) return W4A16_E_WORKSPACE;
    if ((long long)batch * gx > kMaxTickets) return W4A16_E_SHAPE;
    tickets = static_cast<unsigned*>(ws);
    partial = reinterpret_cast<f32x4*>(static_cast<unsigned char*>(ws) + kTicketBytes);
  }
  DecodeArgs e;
  e.ws = partial, e.tickets = tickets, e.kb_per_split = kb_per_split, e.nb_max = nb_max + 1;
  e.wstride = a.KBS * 1024, e.sstride = a.NG * 64;
  // every wave of every split must own at least two blocks (the shortest split is the last one)
  const int last_split = a.KB - (splits - 1) * kb_per_split;
  e.steal = (W >= 8 && last_split >= 2 * W) ? 1 : 0;
  e.splits = splits, e.gx = gx;
  void (*kern)(const GemmArgs, const DecodeArgs) = nullptr;
  const int szs = a.G % 128 == 0 ? 4 : a.G == 64 ? 2 : 1;
#define W4A16_PICK(SZS_)                                                                         \
  if (szs == SZS_) {                                                                             \
    if (act) {                                                                                   \
      if constexpr (kAct) kern = xlds ? gemm_decode_kernel<T, W, DB, SZS_, true, true>           \
                                      : gemm_decode_kernel<T, W, DB, SZS_, true, false>;         \
    } else {                                                                                     \
      if constexpr (kPlain) kern = xlds ? gemm_decode_kernel<T, W, DB, SZS_, false, true>        \
                                        : gemm_decode_kernel<T, W, DB, SZS_, false, false>;      \
    }                                                                                            \
  }
  W4A16_PICK(4) W4A16_PICK(2) W4A16_PICK(1)
#undef W4A16_PICK
  if (!kern) return W4A16_E_SHAPE;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      g_last_hip_error = (int)e;
      return W4A16_E_LAUNCH;
    }
  }
  hipLaunchKernelGGL(kern, dim3(gx, splits, batch), dim3(64 * W), lds, s, a, e);
  return launched();
}

// (t, waves) of the decode kernel; the fused activation takes the even t
#define W4A16_FOR_EACH_DECODE(X) \
  X(1, 16) X(2, 16) X(3, 16) X(1, 8) X(2, 8) X(3, 8) X(4, 8) X(6, 8) X(8, 8) X(2, 4) X(4, 4)

// `depth`: blocks a wave keeps in flight (1 or 2; 0 = the default for the shape)
int dispatch_decode(const GemmArgs& a, int t, int waves, hipStream_t s, int batch, bool act, int splits, void* ws,
                    size_t ws_bytes, int depth = 0) {
  // Two blocks per wave in flight where one would leave the CU short of outstanding requests (8192 x 1280 at
  // 1 row 5.2 us against 5.8 with one block, 11008 x 4096 7.0 against 7.9, 8192 x 3584 5.6 against 6.9), but no
  // more than a CU takes in at a time, ~64 KiB (beyond it the youngest waves' first requests wait for the oldest
  // waves' second blocks; 4096 x 11008 with 3 x 16: 7.0 against 7.4).  That is: up to 16 tile streams per
  // workgroup always, up to 24 from K = 6144 on (3 tiles x 8 waves: 4096 x 11008 7.2 us with one
  // block against 7.45 with two, 8192 x 10240 11.0 against 10.4); with 32 streams and more, one block each is
  // already past what a CU keeps in flight (8192 x 8192 with 2 x 16: 8.85 against 9.16; 4096 x 14336 with
  // 4 x 8: 8.5 against 9.5).
  // 2 tiles x 16 waves stay at two blocks from 5 rows on (fragments from L2: 8192 x 8192 at 8 rows 9.6 against
  // 10.1) and from K = 12800 on (13824 x 5120: 10.6 against 10.9).
  if (depth == 0) {
    const int streams = t * waves;
    depth = (streams < 24 || (streams < 32 && a.K >= 6144) ||
             (streams == 32 && waves == 16 && (a.M > kDecodeMaxRows || a.K >= 12800))) ? 2 : 1;
    // 4 tiles x 4 waves (several workgroups to a CU): one block up to K = 6144 (4096 x 28672 at 1 row 14.0 us
    // against 14.9), two beyond (8 x 14336 x 4096: 46.6 against 52.2)
    if (t == 4 && waves == 4 && a.K < 6144) depth = 1;
  }
#define X(T_, W_)                                                                                     \
  if (t == T_ && waves == W_)                                                                         \
    return depth == 1 ? launch_decode<T_, W_, 1>(a, s, batch, act, splits, ws, ws_bytes)              \
                      : launch_decode<T_, W_, 2>(a, s, batch, act, splits, ws, ws_bytes);
  W4A16_FOR_EACH_DECODE(X)
#undef X
  return W4A16_E_SHAPE;
}

bool decode_built(int t, int waves, bool act) {
  if (act && t % 2 != 0) return false;
#define X(T_, W_) \
  if (t == T_ && waves == W_) return true;
  W4A16_FOR_EACH_DECODE(X)
#undef X
  return false;
}

int pow2_floor(int v) {
  int p = 1;
  while (p * 2 <= v) p *= 2;
  return p;
}

// Default launch shape.
//
// Workgroups own whole columns of the output (no cross-workgroup reduction), so the aim is about
// one workgroup per CU with every CU busy.  Up to 32 rows the streaming kernel runs (x fragments
// straight from L2 into registers); above that the LDS-staged kernel, whose candidates are ranked
// by a small cost model: rounds of 256 workgroups x tiles per workgroup, penalised for 2-row-tile
// shapes (the int4 -> fp16 conversion of a weight fragment is then shared by 2 instead of 4
// matrix instructions) and for half-empty workgroups.
struct Choice {
  bool tiled;
  int mt, t, a, b;  // streaming: a = waves; tiled: a = nw, b = kw
  int splits = 0;   // > 0: the decode kernel with t tiles, a waves and this many K splits
};

// Launch shape of the two 32x32x16 kernels: the built shape with the lowest estimated time,
//   ceil(workgroups / 256 CUs) x (K / 128 x a + b)   microseconds,
// a (per 128-k block) and b (prologue, reduction, stores) per shape, fitted by least squares on all fifteen
// shapes x 14 layer shapes (Llama-2-7B, Llama-2-70B as one of 8 ranks, Mixtral, 2/4/8-way shards of the
// benchmark layer) x 24-4096 rows (scripts/exp_shared_sweep.py): mean loss against the best built shape 1.1 %,
// worst 12 %; with each layer shape held out of the fit in turn, 1.1 % and 12 %.  The private-slab kernel splits K
// over the waves of a workgroup, so it keeps many workgroups on narrow layers at any row count (8192 x 1280 at
// 512 rows: 20 us against 58 for the best shared-slab shape); the shared-slab kernel wins where rows x columns
// alone fill the chip.
Choice choose_mfma32(int M, int K, int N, int batch) {
  struct Shape {
    int mt, t, nw, kw;  // private slabs: nw = 1; shared slab: kw = 1
    double a, b;
  };
  static const Shape shapes[] = {
      {1, 2, 1, 8, 0.1730, 1.378}, {1, 3, 1, 4, 0.2506, 1.434}, {1, 3, 1, 8, 0.2341, 3.025}, {2, 1, 1, 8, 0.1913, 2.045},
      {2, 2, 1, 4, 0.2547, 2.210}, {2, 2, 1, 8, 0.2444, 3.492}, {2, 3, 1, 4, 0.3181, 4.032}, {2, 3, 1, 8, 0.3005, 4.867},
      {2, 1, 8, 1, 0.8776, 4.211}, {2, 2, 8, 1, 1.4627, 6.527}, {2, 3, 8, 1, 2.0270, 8.208}, {4, 1, 8, 1, 1.2832, 6.042},
      {4, 2, 4, 1, 1.3471, 6.449}, {4, 2, 8, 1, 2.1858, 8.772}, {4, 3, 4, 1, 1.8948, 8.294}};
  Choice best{true, 2, 3, 1, -4};
  double best_cost = 1e300;
  for (const Shape& c : shapes) {
    if (c.kw == 1 && K < 256) continue;  // the shared-slab kernel needs two blocks
    const long long wgs = (long long)ceil_div(ceil_div(N, 32), c.t * c.nw) * ceil_div(M, 32 * c.mt) * batch;
    const double cost = (double)((wgs + 255) / 256) * (K / 128.0 * c.a + c.b);
    if (cost < best_cost) {
      best_cost = cost;
      best = c.kw == 1 ? Choice{true, c.mt, c.t, c.nw, -1} : Choice{true, c.mt, c.t, 1, -c.kw};
    }
  }
  return best;
}

Choice choose_cfg(int M, int K, int N, int batch = 1, bool act = false) {
  const int NT = ceil_div(N, 16), KB = ceil_div(K, 128);
  // The 32x32x16 kernels read whole 128-k blocks through 32-bit byte offsets.  (With the fused activation a
  // tile is 16 gate + 16 up columns: as many tiles, and the same launch shapes, as the plain GEMM on N columns.)
  const bool wide_ok = K % 128 == 0 && (long long)NT * padded_kblocks(K) * 1024 < (1LL << 32) &&
                       (long long)M * K < (1LL << 31);
  // From 33 rows: 10-40 % faster than the 16x16x32 kernels on every layer shape measured, and ahead of
  // dequantise + dense GEMM at every row count.
  if (wide_ok && M > 32) return choose_mfma32(M, K, N, batch);
  // 21-32 rows of a single problem: 5-25 % over the 16x16x32 kernels from K = 2048 up (at 24 / 32 rows,
  // 4096 x 11008: 10.2 / 10.6 against 11.6 / 11.6 us; 4096 x 4096: 7.8 / 7.9 against 9.5 / 9.8; 4096 x 14336:
  // 11.0 / 11.4 against 14.6 / 14.8; 8192 x 1280: 12.0 against 15.4; 4096 x 1376: 7.7 / 7.8 against 9.2 / 10.6).
  // At 17 rows the streaming kernel, whose cost grows with the rows, is still ahead on narrow layers and long K
  // (11008 x 4096: 16.4 against 18.9), and at K = 1024 always.
  if (wide_ok && batch == 1 && M > W4A16_WIDE_FROM && K >= 2048) return choose_mfma32(M, K, N, batch);
  const int mtiles = ceil_div(M, 16);
  const int cus = 256;
  // 24-32 rows: the LDS-staged kernel's 2-row-tile shapes beat the streaming kernel by ~12 % where K
  // is long or the layer wide enough to amortise their prologue (11.9 vs 13.5 us on 4096 x 11008,
  // 20.7 vs 23.9 on 11008 x 4096; on 4096 x 4096 from 30 rows only: 9.8 vs 10.3 at 32): x slabs arrive
  // in whole cache lines instead of 64-byte fragments.  With fewer rows the streaming kernel's fragment loads touch fewer lines and
  // it stays ahead.  Very wide layers (22016 columns) share each slab between two waves.
  const bool wide = (long long)NT * batch > 3 * cus;
  const bool two_tiles_staged = mtiles == 2 && (wide || (M >= 24 && (KB >= 64 || NT >= 512)) || (M >= 30 && KB >= 32 && NT >= 256)) && KB >= 16 &&
                                (!act || wide);  // fused activation: built for even NW only
  if (mtiles <= 2 && !two_tiles_staged) {
    const int MT = mtiles;
    int Wv = (MT == 1) ? 16 : 8;
    while (Wv > 4 && Wv > pow2_floor(KB)) Wv /= 2;
    // one or two rows over a very long K (14336): 8 waves with four blocks each in flight beat 16
    // (10.2 vs 10.4 us); up to K = 11008 it is the other way round (8.5 vs 8.8 us)
    if (M <= 2 && KB >= 100 && Wv == 16) Wv = 8;
    // Many tiles (batched launches, very wide layers): workgroups run back to back on a CU, and
    // small ones, several to a CU, overlap one's loads with another's arithmetic and reduction -
    // 5.4 TB/s of weights at 1-16 rows against 3.9 for one 16-wave workgroup per CU.
    const long long tiles = (long long)NT * batch;
    if (MT == 1 && KB >= 8) {
      // From six tiles per CU on, the more rows the more tiles per workgroup (a wave's x fragments then
      // serve more of them): 8192 x 57344 with the fused activation 46 / 51 / 53 us at 1 / 8 / 16 rows,
      // against 50 / 54 / 70 with two tiles and eight waves throughout.
      if (act && tiles >= 6 * cus && M > 8) return Choice{false, 1, 8, 8, 0};  // measured with the fused activation only
      if (tiles >= 16 * cus || (tiles >= 6 * cus && M > 4)) return Choice{false, 1, 4, 4, 0};
      if (tiles >= 6 * cus) return Choice{false, 1, 2, 4, 0};
      // More than three tiles per CU (a 22016-wide gate_up layer): 16-wave workgroups of 3 tiles would
      // run in two uneven rounds; 8-wave workgroups are all resident at once (12.6 vs 14.3 us at
      // 1 row, 13.8 vs 16.3 us at 16 rows on 4096 x 22016).
      // Among 2, 3, 4 and 6 tiles per workgroup the one that leaves the busiest CU the fewest tiles
      // (896 tiles, a 14336-wide layer: 4 with T = 2 or 4, 6 with T = 3; 9.2 vs 11.8 us at 1 row); few
      // rows prefer small workgroups, 9-16 rows large ones (their x fragments serve more tiles).
      if (!act && tiles > 3 * cus) {
        static const int few[4] = {3, 2, 4, 6}, many[4] = {6, 4, 3, 2};
        const int* order = M <= 8 ? few : many;
        int best_t = order[0];
        long long best_load = -1;
        for (int i = 0; i < 4; ++i) {
          const long long load = ceil_div(ceil_div(tiles, order[i]), cus) * order[i];
          if (best_load < 0 || load < best_load) best_load = load, best_t = order[i];
        }
        return Choice{false, 1, best_t, 8, 0};
      }
    }
    if (act) {
      // gate/up tile pairs: an even T from the fused-activation instantiations
      int Tt = 2 * ceil_div(tiles / 2, cus);
      if (MT == 1 && Tt == 2 && Wv == 16) return Choice{false, 1, 2, 16, 0};
      if (Wv > 8) Wv = 8;
      const int tmax = (MT == 1) ? (Wv == 8 ? 8 : 4) : (Wv == 8 ? 6 : 2);
      if (Tt > tmax) Tt = tmax;
      return Choice{false, MT, Tt, Wv, 0};
    }
    int Tt = ceil_div(tiles, cus);  // one workgroup per CU (registers and LDS admit no second one)
    if (Tt < 1) Tt = 1;
    if (Tt == 5) Tt = 6;
    if (Tt == 7) Tt = 8;
    // largest T instantiated for this (MT, waves): see W4A16_FOR_EACH_CFG
    int tmax;
    if (MT == 1) tmax = (Wv == 16) ? 3 : (Wv == 8) ? 8 : 4;
    else tmax = (Wv == 16) ? 1 : (Wv == 8) ? 6 : 3;
    if (Tt > tmax) Tt = tmax;
    return Choice{false, MT, Tt, Wv, 0};
  }
  // 65-128 rows on a layer of at most 256 column tiles (4096 columns): 64 x 4 = 256 workgroups of the
  // streaming kernel (2 row tiles x 4 column tiles, 8 waves) fill the chip exactly and beat the
  // LDS-staged shapes by ~10 % (13.6 vs 15.0 us on 4096 x 4096, 30.2 vs 33.3 on 11008 x 4096).
  // 33-64 rows on a layer of 448-512 column tiles: two row-blocks of the same shape fill the chip in one round
  // (1024 x 8192: 5.5-5.9 us against 6.4-6.6 for the best LDS-staged shape; 8192 x 7168: 23.2 against 24.4).
  if (!act && mtiles >= 3 && mtiles <= 4 && KB >= 8) {
    const long long wgs = (long long)ceil_div((long long)NT * batch, 4) * 2;
    if (wgs <= cus && wgs * 8 >= cus * 7) return Choice{false, 2, 4, 8, 0};
  }
  if (!act && mtiles > 4 && mtiles <= 8 && (long long)NT * batch <= cus && (long long)NT * batch * 8 >= cus * 7 && KB >= 16)
    return Choice{false, 2, 4, 8, 0};
  if (two_tiles_staged) {
    if (wide) return Choice{true, 2, 3, 2, 4};
    return Choice{true, 2, 3, 1, padded_kblocks(K) % 8 == 0 ? 8 : 4};
  }
  static const int cand[][4] = {{4, 3, 2, 4}, {4, 3, 4, 2}, {4, 3, 1, 4}, {4, 2, 2, 4},
                                {2, 3, 2, 4}, {2, 3, 1, 4}, {2, 3, 1, 8}};
  Choice best{true, 4, 3, 2, 4};
  double best_cost = 1e300;
  for (const auto& c : cand) {
    const int mt = c[0], t = c[1], nw = c[2], kw = c[3];
    if (mtiles <= 4 && mt == 4 && mtiles <= 2) continue;
    if (act && (nw % 2 || t != 3)) continue;  // fused-activation instantiations: see kActTiled
    if (padded_kblocks(K) % kw) continue;
    const long long ctas = (long long)ceil_div(NT, nw * t) * ceil_div(mtiles, mt) * batch;
    // whole rounds of workgroups while they are few; with many rounds the partial last one fades
    const double rounds = ctas > 2 * cus ? (double)ctas / cus + 0.5 : (double)ceil_div(ctas, cus);
    double work = mt * t * nw;                        // matrix work per workgroup, in output tiles
    work *= (mt == 4) ? 1.0 : 1.45;                   // conversion shared by fewer MFMAs
    work *= (t == 3) ? 1.0 : 1.15;                    // x fragments re-used by 2 instead of 3 column tiles
    work *= (nw * kw == 8) ? 1.0 : 1.4;               // 4-wave workgroups: one wave per SIMD
    work *= 1.0 + 0.6 / nw;                           // x slab staged once per nw column waves
    // every round also pays a prologue, the cross-K-group reduction and the stores: about a third of a
    // 24-tile workgroup's time.  Fitted on 11 layer shapes x 48-512 rows (scripts/exp_tiled_sweep.py):
    // mean loss against the best built shape 1.5 % (2.4 % without this term).
    const double cost = rounds * (work + 8.0);
    if (cost < best_cost) {
      best_cost = cost;
      best = Choice{true, mt, t, nw, kw};
    }
  }
  return best;
}

// The decode kernel takes over the streaming kernel's launches of up to 16 rows wherever it is built for the
// shape chosen (4-7 % faster at 5-16 rows, 8-20 % at up to 4).  With a workspace a launch that would leave most CUs idle (few column tiles, long K) is split
// over K as well.
Choice choose_with_decode(int M, int K, int N, int G, int batch, bool act, bool have_ws) {
  Choice c = choose_cfg(M, K, N, batch, act);
  if (c.tiled || c.mt != 1 || M > kDecodeMaxRowsL2 || !W4A16_DECODE) return c;
  GemmArgs probe;
  probe.M = M, probe.K = K, probe.N = N, probe.G = G;
  probe.NT = ceil_div(N, 16), probe.KB = ceil_div(K, 128), probe.KBS = padded_kblocks(K), probe.NG = K / G;
  if (!decode_ok(probe)) return c;
  // One or two rows over a very long K: the streaming kernel prefers 8 waves there, the decode kernel 16 with two
  // blocks each in flight (14336 x 4096 at 1 row 8.6 us against 9.9 with 8 waves and 10.2 for the streaming kernel).
  const long long col_tiles = (long long)(act ? probe.NT / 2 : probe.NT) * batch;
  if (!act && c.a == 8 && M <= 2 && probe.KB >= 100 && col_tiles <= 3 * 256) {
    const int t16 = (int)(col_tiles <= 256 ? 1 : col_tiles <= 512 ? 2 : 3);
    if (decode_lds_bytes(t16, 16, ceil_div(probe.KB, 16) + 1, M, G) <= 160 * 1024) c.t = t16, c.a = 16;
  }
  // Three to six tiles per CU: the streaming kernel's order of preference (small workgroups
  // first) turns round, one workgroup per CU with the fewest tiles on the busiest CU is ahead (5120 x 13824:
  // 9.35 us with 4 tiles against 10.1 with 2 and 10.4 with 3; 4096 x 14336: 8.5 against 9.3; 4096 x 22016:
  // 12.0 with 6 against 12.1 with 3).
  if (!act && c.a == 8 && col_tiles > 3 * 256 && col_tiles < 6 * 256) {
    static const int order[4] = {6, 4, 3, 2};
    long long best = -1;
    for (int i = 0; i < 4; ++i) {
      const long long load = ceil_div(ceil_div(col_tiles, order[i]), 256) * order[i];
      if (best < 0 || load < best) best = load, c.t = order[i];
    }
  }
  // From six tiles per CU on (very wide layers, batched launches): 4-wave workgroups of 4 tiles, several to a CU,
  // at every row count; the streaming kernel's preferences (2 tiles up to 4 rows, 8 x 8 from 9 rows with the fused
  // activation) do not carry over: 4096 x 28672 at 1 row 14.0 us against 15.0 with 2 tiles, with the fused
  // activation 13.2 against 14.9; 8 x (14336 x 4096) at 1 row 46.6 against over 56; 8 x (4096 x 28672) with the
  // fused activation at 16 rows 97 against 106 for 8 x 8.
  if ((long long)probe.NT * batch >= 6 * 256 && ((c.a == 4 && (c.t == 2 || c.t == 4)) || (act && c.a == 8 && c.t == 8)))
    c.t = 4, c.a = 4;
  if (!decode_built(c.t, c.a, act)) return c;
  const int per_wg = act ? c.t / 2 : c.t;
  const int nb_max = ceil_div(probe.KB, c.a);
  if (decode_lds_bytes(c.t, c.a, nb_max + 1, M, G) > 160 * 1024) return c;
  c.splits = 1;
  if (have_ws && M <= kDecodeMaxRows) {
    // A split pays where a wave's chain of blocks is long (its requests go out a block or two at a time, a
    // memory latency each) and the workgroups are few: from K = 12800 on, about 40 blocks per split
    // (14336 x 512 at 1 row 7.4 -> 6.0 us with 3 splits, 16384 x 1024 8.1 -> 6.7; at K = 11008 it is a wash and
    // at K <= 8192 the exchange through memory costs more than it saves).
    const long long wgs = (long long)ceil_div(act ? probe.NT / 2 : probe.NT, per_wg) * batch;
    if (probe.KB >= 100 && wgs <= 128) {
      int sp = (probe.KB + 20) / 40;
      if (sp > 256 / wgs) sp = (int)(256 / wgs);
      if (sp > kMaxSplits) sp = kMaxSplits;
      if (sp > 1) c.splits = sp;
    }
  }
  return c;
}

// ---- the library's own scratch, for callers that cannot hand in a workspace they allocated ----
// K-split launches need ticket counters and room for partial sums (see gemm_decode_kernel).  This array is that
// room for w4a16_gemm_packed_f16_decode_static: every GPU the library runs on holds its own copy, at its own
// address, from the moment the code object is loaded until the process ends.  In host code the array's name is
// only a handle for the runtime: the device address comes from hipGetSymbolAddress, once per device, and that
// pointer is what the kernels get.
constexpr size_t kStaticScratchBytes = (size_t)1 << 20;  // 64 KiB of counters + 960 KiB of partial sums
}  // namespace (the symbol below has to be visible to HIP_SYMBOL)

static __device__ __align__(256) unsigned char scratch[(size_t)1 << 20];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[kMaxDevices] = {};

// The current device's copy of `scratch` (looked up on the first call on that device).
int static_scratch(void** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    g_last_hip_error = (int)e;
    return W4A16_E_LAUNCH;
  }
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[dev] == nullptr) {
    void* ptr = nullptr;
    e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || ptr == nullptr) {
      g_last_hip_error = (int)e;
      return W4A16_E_LAUNCH;
    }
    // The counters must start at zero: the loader zero-fills the array, and every kernel puts the counters it
    // used back to zero before it ends, so nothing is cleared here (a clear enqueued on this call's stream
    // could run beside a launch that another stream already has on the array).
    g_scratch_ptr[dev] = ptr;
  }
  *out = g_scratch_ptr[dev];
  return W4A16_OK;
}

}  // namespace

extern "C" {

const char* w4a16_strerror(int code) {
  switch (code) {
    case W4A16_OK: return "ok";
    case W4A16_E_OC_PACK: return "OC is not multiple of pack_num = 8";
    case W4A16_E_GROUP32: return "Group size should be a multiple of 32";
    case W4A16_E_K_GROUP: return "IC is not multiple of Group size";
    case W4A16_E_NULL: return "null pointer argument";
    case W4A16_E_SHAPE: return "invalid shape or launch configuration";
    case W4A16_E_ALIGN: return "pointer is not 16-byte aligned";
    case W4A16_E_LAUNCH: return "HIP launch failed";
    case W4A16_E_TOO_LARGE: return "tensor too large for 32-bit indexing";
    case W4A16_E_WORKSPACE: return "workspace too small (or the launch needs more than the library's own scratch)";
    default: return "unknown w4a16 error";
  }
}

int w4a16_last_hip_error(void) { return g_last_hip_error; }

// For the library's other source files (not exported): records the hipError_t of a failed launch.
__attribute__((visibility("hidden"))) void w4a16_note_hip_error(int e) { g_last_hip_error = e; }

int w4a16_abi_version(void) { return W4A16_ABI_VERSION; }

#ifdef W4A16_STAMPS
// Diagnostic build only: copies the stamp buffer to host memory (synchronises).
int w4a16_dbg_stamps(void* host_out, size_t bytes) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), bytes, 0, hipMemcpyDeviceToHost);
}
#endif

size_t w4a16_packed_weight_bytes(int K, int N) {
  if (K <= 0 || N <= 0) return 0;
  return (size_t)ceil_div(N, 16) * padded_kblocks(K) * 1024;
}

size_t w4a16_packed_sz_bytes(int K, int N, int G) {
  if (K <= 0 || N <= 0 || G <= 0 || K % G) return 0;
  return (size_t)ceil_div(N, 16) * (K / G) * 16 * 4;
}

int w4a16_gemm_f16(const void* x, const void* qweight, const void* scales, const void* qzeros,
                   void* out, int M, int K, int N, int G, int split_k_iters, void* stream) {
  (void)split_k_iters;
  if (!x || !qweight || !scales || !qzeros || !out) return W4A16_E_NULL;
  if (M <= 0) return W4A16_E_SHAPE;
  const int rc = check_common(K, N, G);
  if (rc) return rc;
  if ((long long)M * N >= (1LL << 31) || (long long)M * K >= (1LL << 31)) return W4A16_E_TOO_LARGE;
  if (misaligned(scales) || misaligned(out)) return W4A16_E_ALIGN;
  clear_stale_error();
  const hipStream_t s = static_cast<hipStream_t>(stream);
  const int gx = ceil_div(N / 8, 16);
#define W4A16_STOCK(MT_, KW_)                                                                              \
  do {                                                                                                     \
    constexpr int lds = (KW_ / 2) * 8 * MT_ * 1024;                                                        \
    static_assert(lds <= 160 * 1024, "LDS budget");                                                        \
    if (lds > 64 * 1024) {                                                                                 \
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_stock_kernel<MT_, KW_>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);           \
      if (e != hipSuccess) {                                                                               \
        g_last_hip_error = (int)e;                                                                         \
        return W4A16_E_LAUNCH;                                                                             \
      }                                                                                                    \
    }                                                                                                      \
    hipLaunchKernelGGL((gemm_stock_kernel<MT_, KW_>), dim3(gx, ceil_div(M, 16 * MT_)), dim3(KW_ * 64), lds, s, \
                       (const f16*)x, (const uint32_t*)qweight, (const f16*)scales, (const uint32_t*)qzeros,  \
                       (f16*)out, M, K, N, G);                                                             \
  } while (0)
  if (M <= 16)
    W4A16_STOCK(1, 16);
  else if (M <= 32)
    W4A16_STOCK(2, 8);
  else
    W4A16_STOCK(4, 8);
#undef W4A16_STOCK
  return launched();
}

int w4a16_dequantize_f16(const void* qweight, const void* scales, const void* qzeros, void* out,
                         int K, int N, int G, void* stream) {
  if (!qweight || !scales || !qzeros || !out) return W4A16_E_NULL;
  const int rc = check_common(K, N, G);
  if (rc) return rc;
  if (misaligned(scales) || misaligned(out)) return W4A16_E_ALIGN;
  clear_stale_error();
  const size_t total = (size_t)K * (N / 8);
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(dequant_stock_kernel, dim3(blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), (const uint32_t*)qweight,
                     (const f16*)scales, (const uint32_t*)qzeros, (f16*)out, K, N, G);
  return launched();
}

int w4a16_repack(const void* qweight, const void* scales, const void* qzeros, void* packed_w,
                 void* packed_sz, int K, int N, int G, void* stream) {
  if (!qweight || !scales || !qzeros || !packed_w || !packed_sz) return W4A16_E_NULL;
  const int rc = check_common(K, N, G);
  if (rc) return rc;
  if (K % 32) return W4A16_E_SHAPE;
  if (misaligned(packed_w) || misaligned(packed_sz)) return W4A16_E_ALIGN;
  clear_stale_error();
  const int NT = ceil_div(N, 16), KB = padded_kblocks(K), NG = K / G;
  const hipStream_t s = static_cast<hipStream_t>(stream);
  {
    const size_t total = (size_t)NT * KB * 256;
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(repack_weight_kernel, dim3(blocks), dim3(256), 0, s,
                       (const uint32_t*)qweight, (const uint32_t*)qzeros, (uint32_t*)packed_w, K,
                       N, G, NT, KB);
    const int r = launched();
    if (r) return r;
  }
  {
    const size_t total = (size_t)NT * NG * 16;
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(repack_sz_kernel, dim3(blocks), dim3(256), 0, s, (const f16*)scales,
                       (const uint32_t*)qzeros, (uint32_t*)packed_sz, N, NT, NG);
  }
  return launched();
}

int w4a16_gemm_packed_f16_cfg(const void* x, const void* packed_w, const void* packed_sz,
                              const void* bias, void* out, int M, int K, int N, int G, int mt,
                              int t, int waves, void* stream) {
  GemmArgs a;
  const int rc = fill_args(&a, x, packed_w, packed_sz, bias, out, M, K, N, G);
  if (rc) return rc;
  clear_stale_error();
  return dispatch_packed(a, mt, t, waves, static_cast<hipStream_t>(stream));
}

int w4a16_gemm_packed_f16_tiled_cfg(const void* x, const void* packed_w, const void* packed_sz,
                                    const void* bias, void* out, int M, int K, int N, int G, int mt,
                                    int t, int nw, int kw, void* stream) {
  GemmArgs a;
  const int rc = fill_args(&a, x, packed_w, packed_sz, bias, out, M, K, N, G);
  if (rc) return rc;
  clear_stale_error();
  return dispatch_tiled(a, mt, t, nw, kw, static_cast<hipStream_t>(stream));
}

int w4a16_gemm_packed_f16_wide_cfg(const void* x, const void* packed_w, const void* packed_sz,
                                   const void* bias, void* out, int M, int K, int N, int G, int mt,
                                   int t, int kw, void* stream) {
  GemmArgs a;
  const int rc = fill_args(&a, x, packed_w, packed_sz, bias, out, M, K, N, G);
  if (rc) return rc;
  clear_stale_error();
  if (kw < 0) return dispatch_wide_shared(a, mt, t, -kw, static_cast<hipStream_t>(stream));
  return dispatch_wide(a, mt, t, kw, static_cast<hipStream_t>(stream));
}

// Shared body of the batched / fused-epilogue entry points.  mt == 0: pick the launch shape.
static int gemm_packed_general(const void* x, long long x_batch_stride, const void* packed_w,
                               const void* packed_sz, const void* bias, void* out,
                               long long out_batch_stride, int M, int K, int N, int G, int batch,
                               int epilogue, int mt, int t, int a_, int b_, void* stream,
                               const void* problem_mask = nullptr, int active_hint = 0, void* ws = nullptr,
                               size_t ws_bytes = 0) {
  if (M <= 0 || K <= 0 || N <= 0 || batch <= 0) return W4A16_E_SHAPE;
  if (problem_mask && (reinterpret_cast<uintptr_t>(problem_mask) & 3u)) return W4A16_E_ALIGN;
  if (epilogue != W4A16_EPI_NONE && epilogue != W4A16_EPI_SILU_MUL && epilogue != W4A16_EPI_ADD_TO_OUT)
    return W4A16_E_SHAPE;
  const bool act = epilogue == W4A16_EPI_SILU_MUL;
  if (act && N % 32) return W4A16_E_SHAPE;  // two halves of whole 16-column tiles
  const long long out_cols = act ? N / 2 : N;
  GemmArgs a;
  // a single problem needs no strides; give fill_args consistent ones anyway
  const int rc = fill_args(&a, x, packed_w, packed_sz, bias, out, M, K, N, G, batch, x_batch_stride,
                           batch == 1 ? (long long)M * N : out_batch_stride);
  if (rc) return rc;
  if (batch != 1) {
    if (out_batch_stride < M * out_cols || out_batch_stride % 4) return W4A16_E_SHAPE;
    a.out_bs = out_batch_stride;
  }
  if ((long long)batch * out_batch_stride >= (1LL << 40)) return W4A16_E_TOO_LARGE;
  a.add_out = epilogue == W4A16_EPI_ADD_TO_OUT;
  a.mask = static_cast<const int*>(problem_mask);
  clear_stale_error();
  const hipStream_t s = static_cast<hipStream_t>(stream);
  if (mt == 0) {
    // with a mask the launch shape is chosen for the problems expected to be on: the others cost nothing
    const int busy = (problem_mask && active_hint > 0 && active_hint < batch) ? active_hint : batch;
    const Choice c = choose_with_decode(M, K, N, G, busy, act, ws != nullptr);
    mt = c.mt, t = c.t, a_ = c.a, b_ = c.tiled ? c.b : 0;
    if (c.splits > 0) mt = -1, b_ = c.splits;
  }
  if (mt == -1)  // decode kernel: (t, waves, splits + 256 * depth)
    return dispatch_decode(a, t, a_, s, batch, act, b_ & 255, ws, ws_bytes, b_ >> 8);
  if (b_ < 0) {  // 32x32x16 kernels: (mt, t, 1, -kw) private slabs, (mt, t, nw > 1, any b < 0) shared slab
    return a_ > 1 ? dispatch_wide_shared(a, mt, t, a_, s, batch, act) : dispatch_wide(a, mt, t, -b_, s, batch, act);
  }
  if (b_ > 0) return dispatch_tiled(a, mt, t, a_, b_, s, batch, act);
  return dispatch_packed(a, mt, t, a_, s, batch, act);
}

int w4a16_gemm_packed_f16(const void* x, const void* packed_w, const void* packed_sz,
                          const void* bias, void* out, int M, int K, int N, int G, void* stream) {
  return gemm_packed_general(x, 0, packed_w, packed_sz, bias, out, 0, M, K, N, G, 1, W4A16_EPI_NONE, 0, 0, 0, 0,
                             stream);
}

int w4a16_gemm_packed_f16_batched_cfg(const void* x, long long x_batch_stride, const void* packed_w,
                                      const void* packed_sz, const void* bias, void* out,
                                      long long out_batch_stride, int M, int K, int N, int G, int batch,
                                      int mt, int t, int a, int b, void* stream) {
  if (mt <= 0) return W4A16_E_SHAPE;
  return gemm_packed_general(x, x_batch_stride, packed_w, packed_sz, bias, out, out_batch_stride, M, K,
                             N, G, batch, W4A16_EPI_NONE, mt, t, a, b, stream);
}

int w4a16_gemm_packed_f16_batched(const void* x, long long x_batch_stride, const void* packed_w,
                                  const void* packed_sz, const void* bias, void* out,
                                  long long out_batch_stride, int M, int K, int N, int G, int batch,
                                  void* stream) {
  return gemm_packed_general(x, x_batch_stride, packed_w, packed_sz, bias, out, out_batch_stride, M, K,
                             N, G, batch, W4A16_EPI_NONE, 0, 0, 0, 0, stream);
}

int w4a16_gemm_packed_f16_ex(const void* x, long long x_batch_stride, const void* packed_w,
                             const void* packed_sz, const void* bias, void* out,
                             long long out_batch_stride, int M, int K, int N, int G, int batch,
                             int epilogue, int mt, int t, int a, int b, void* stream) {
  if (mt < 0) return W4A16_E_SHAPE;
  return gemm_packed_general(x, x_batch_stride, packed_w, packed_sz, bias, out, out_batch_stride, M, K,
                             N, G, batch, epilogue, mt, t, a, b, stream);
}

int w4a16_gemm_packed_f16_masked(const void* x, long long x_batch_stride, const void* packed_w,
                                 const void* packed_sz, const void* bias, void* out,
                                 long long out_batch_stride, int M, int K, int N, int G, int batch,
                                 int epilogue, const void* problem_mask, int active_hint, void* stream) {
  if (active_hint < 0) return W4A16_E_SHAPE;
  return gemm_packed_general(x, x_batch_stride, packed_w, packed_sz, bias, out, out_batch_stride, M, K,
                             N, G, batch, epilogue, 0, 0, 0, 0, stream, problem_mask, active_hint);
}

int w4a16_gemm_packed_f16_decode_cfg(const void* x, long long x_batch_stride, const void* packed_w,
                                     const void* packed_sz, const void* bias, void* out,
                                     long long out_batch_stride, int M, int K, int N, int G, int batch,
                                     int epilogue, int t, int waves, int depth, int splits, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  if (t <= 0 || waves <= 0 || splits <= 0 || splits > kMaxSplits || depth < 0 || depth > 2) return W4A16_E_SHAPE;
  return gemm_packed_general(x, x_batch_stride, packed_w, packed_sz, bias, out, out_batch_stride, M, K, N, G,
                             batch, epilogue, -1, t, waves, splits + 256 * depth, stream, nullptr, 0, workspace,
                             workspace_bytes);
}

size_t w4a16_decode_workspace_bytes(int K, int N, int batch, int epilogue, int t, int waves, int splits) {
  if (K <= 0 || N <= 0 || batch <= 0 || t <= 0 || waves <= 0 || splits <= 1 || splits > kMaxSplits) return 0;
  const bool act = epilogue == W4A16_EPI_SILU_MUL;
  if (!decode_built(t, waves, act)) return 0;  // (the fused activation takes tile pairs: no odd t, so t / 2 >= 1 below)
  const int KB = ceil_div(K, 128);
  splits = ceil_div(KB, ceil_div(KB, splits));
  const int NT = ceil_div(N, 16);
  return decode_ws_bytes(batch, ceil_div(act ? NT / 2 : NT, act ? t / 2 : t), splits, act);
}

size_t w4a16_workspace_bytes(int M, int K, int N, int G, int batch, int epilogue) {
  if (M <= 0 || K <= 0 || N <= 0 || batch <= 0 || check_common(K, N, G)) return 0;
  const bool act = epilogue == W4A16_EPI_SILU_MUL;
  const Choice c = choose_with_decode(M, K, N, G, batch, act, true);
  if (c.splits <= 1) return 0;
  return w4a16_decode_workspace_bytes(K, N, batch, epilogue, c.t, c.a, c.splits);
}

int w4a16_gemm_packed_f16_ws(const void* x, long long x_batch_stride, const void* packed_w, const void* packed_sz,
                             const void* bias, void* out, long long out_batch_stride, int M, int K, int N, int G,
                             int batch, int epilogue, void* workspace, size_t workspace_bytes, void* stream) {
  // a workspace too small for the split the library would choose: the unsplit launch, not an error
  if (workspace && workspace_bytes < w4a16_workspace_bytes(M, K, N, G, batch, epilogue)) workspace = nullptr;
  return gemm_packed_general(x, x_batch_stride, packed_w, packed_sz, bias, out, out_batch_stride, M, K, N, G,
                             batch, epilogue, 0, 0, 0, 0, stream, nullptr, 0, workspace, workspace_bytes);
}

size_t w4a16_static_scratch_bytes(void) { return kStaticScratchBytes; }

int w4a16_gemm_packed_f16_decode_static(const void* x, long long x_batch_stride, const void* packed_w,
                                        const void* packed_sz, const void* bias, void* out,
                                        long long out_batch_stride, int M, int K, int N, int G, int batch,
                                        int epilogue, int t, int waves, int depth, int splits, void* stream) {
  if (M <= 0 || K <= 0 || N <= 0 || batch <= 0 || t < 0) return W4A16_E_SHAPE;
  if (check_common(K, N, G)) return check_common(K, N, G);
  if (t == 0) {  // the library's choice, as w4a16_gemm_packed_f16_ws makes it with a workspace
    const Choice c = choose_with_decode(M, K, N, G, batch, epilogue == W4A16_EPI_SILU_MUL, true);
    if (c.splits <= 0) return W4A16_E_SHAPE;  // not a decode-kernel problem: nothing here needs scratch
    t = c.t, waves = c.a, depth = 0, splits = c.splits;
  }
  // the size is checked against the array before anything else is done with it
  if (w4a16_decode_workspace_bytes(K, N, batch, epilogue, t, waves, splits) > kStaticScratchBytes)
    return W4A16_E_WORKSPACE;
  void* ptr = nullptr;
  if (splits > 1) {
    const int rc = static_scratch(&ptr);
    if (rc) return rc;
  }
  return w4a16_gemm_packed_f16_decode_cfg(x, x_batch_stride, packed_w, packed_sz, bias, out, out_batch_stride, M, K,
                                          N, G, batch, epilogue, t, waves, depth, splits, ptr, kStaticScratchBytes,
                                          stream);
}

int w4a16_launch_shape_ws(int M, int K, int N, int G, int batch, int epilogue, int shape[5]) {
  if (!shape) return W4A16_E_NULL;
  if (M <= 0 || K <= 0 || N <= 0 || batch <= 0 || check_common(K, N, G)) return W4A16_E_SHAPE;
  const Choice c = choose_with_decode(M, K, N, G, batch, epilogue == W4A16_EPI_SILU_MUL, true);
  if (c.splits <= 0) return w4a16_launch_shape(M, K, N, batch, epilogue, shape);
  shape[0] = W4A16_KERNEL_DECODE, shape[1] = 1, shape[2] = c.t, shape[3] = c.a, shape[4] = c.splits;
  return W4A16_OK;
}

int w4a16_launch_shape(int M, int K, int N, int batch, int epilogue, int shape[5]) {
  if (!shape) return W4A16_E_NULL;
  if (M <= 0 || K <= 0 || N <= 0 || batch <= 0) return W4A16_E_SHAPE;
  const Choice c = choose_with_decode(M, K, N, 128, batch, epilogue == W4A16_EPI_SILU_MUL, false);
  shape[1] = c.mt, shape[2] = c.t;
  if (c.splits > 0) {
    shape[0] = W4A16_KERNEL_DECODE, shape[3] = c.a, shape[4] = c.splits;
  } else if (!c.tiled) {
    shape[0] = W4A16_KERNEL_STREAMING, shape[3] = c.a, shape[4] = 0;
  } else if (c.b > 0) {
    shape[0] = W4A16_KERNEL_LDS_STAGED, shape[3] = c.a, shape[4] = c.b;
  } else if (c.a > 1) {
    shape[0] = W4A16_KERNEL_MFMA32_SHARED, shape[3] = c.a, shape[4] = 1;
  } else {
    shape[0] = W4A16_KERNEL_MFMA32, shape[3] = 1, shape[4] = -c.b;
  }
  return W4A16_OK;
}

int w4a16_built_shape(int kernel, int index, int shape[4]) {
  if (!shape) return W4A16_E_NULL;
  if (index < 0) return W4A16_E_SHAPE;
  int i = 0;
  shape[0] = shape[1] = shape[2] = shape[3] = 0;
#define X3(A_, B_, C_) \
  if (i++ == index) return shape[0] = A_, shape[1] = B_, shape[2] = C_, W4A16_OK;
#define X4(A_, B_, C_, D_) \
  if (i++ == index) return shape[0] = A_, shape[1] = B_, shape[2] = C_, shape[3] = D_, W4A16_OK;
#define X2(A_, B_) \
  if (i++ == index) return shape[0] = A_, shape[1] = B_, W4A16_OK;
  switch (kernel) {
    case W4A16_KERNEL_STREAMING: W4A16_FOR_EACH_CFG(X3) break;
    case W4A16_KERNEL_LDS_STAGED: W4A16_FOR_EACH_TILED(X4) break;
    case W4A16_KERNEL_MFMA32: W4A16_FOR_EACH_WIDE(X3) break;
    case W4A16_KERNEL_MFMA32_SHARED: W4A16_FOR_EACH_WIDE_SHARED(X3) break;
    case W4A16_KERNEL_DECODE: W4A16_FOR_EACH_DECODE(X2) break;
    default: break;
  }
#undef X2
#undef X3
#undef X4
  return W4A16_E_SHAPE;
}

int w4a16_stream_read(const void* src, size_t bytes, void* sink, void* stream) {
  if (!src || !sink) return W4A16_E_NULL;
  if (misaligned(src) || bytes < 16) return W4A16_E_ALIGN;
  clear_stale_error();
  hipLaunchKernelGGL(stream_read_kernel, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const u32x4*>(src), bytes / 16, static_cast<uint32_t*>(sink));
  return launched();
}

int w4a16_dequantize_packed_f16(const void* packed_w, const void* packed_sz, void* out, int K,
                                int N, int G, void* stream) {
  if (!packed_w || !packed_sz || !out) return W4A16_E_NULL;
  const int rc = check_common(K, N, G);
  if (rc) return rc;
  clear_stale_error();
  const int NT = ceil_div(N, 16), KB = padded_kblocks(K), NG = K / G;
  if (misaligned(packed_w) || misaligned(out)) return W4A16_E_ALIGN;
  // x: groups of 8 column tiles; y: enough 4-wave workgroups over the k blocks for ~8 per CU
  const int groups = ceil_div(NT, 8);
  int ky = ceil_div(2048, groups);
  if (ky > ceil_div(KB, 4)) ky = ceil_div(KB, 4);
  if (ky < 1) ky = 1;
  hipLaunchKernelGGL(dequant_packed_kernel, dim3(groups, ky), dim3(256), 0,
                     static_cast<hipStream_t>(stream), (const uint32_t*)packed_w,
                     (const uint32_t*)packed_sz, (f16*)out, K, N, G, NT, KB, NG);
  return launched();
}

}  // extern "C"
