// Shared by the translation units of libw4a16.so: types, the bit-exact int4 -> fp16 conversion, the kernel
// argument blocks, the lists of built launch shapes, and the small host helpers of the launchers.
// One translation unit per kernel family (w4a16_streaming.hip, w4a16_decode.hip, w4a16_tiled.hip, w4a16_wide.hip,
// w4a16_wide_shared.hip, w4a16_stock.hip) plus w4a16.hip with the launch choosers and the C ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>


#include "../../include/w4a16.h"

// Compile-time ablation switch for kernel experiments (scripts/): a bit mask whose bits remove one cost each from a
// kernel (results are then wrong by design); 0 in every shipped build.
#ifndef W4A16_DBG
#define W4A16_DBG 0
#endif

#define W4A16_HIDDEN __attribute__((visibility("hidden")))

// Records the hipError_t of a failed launch for w4a16_last_hip_error() (defined in w4a16.hip).
extern "C" W4A16_HIDDEN void w4a16_note_hip_error(int e);

namespace W4A16_HIDDEN w4a16 {

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

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
  // (stage by stage: a packed instruction right behind the one whose result it reads costs a wait state)
  const f16x2 e0 = h0 + c.nz1;
  const f16x2 e1 = h1 * k.sixteenth + c.nz2;
  const f16x2 e2 = h2 + c.nz1;
  const f16x2 e3 = h3 * k.sixteenth + c.nz2;
  const f16x2 d0 = e0 * c.s;
  const f16x2 d1 = e1 * c.s;
  const f16x2 d2 = e2 * c.s;
  const f16x2 d3 = e3 * c.s;
  return f16x8{d0.x, d0.y, d1.x, d1.y, d2.x, d2.y, d3.x, d3.y};
}

__device__ __forceinline__ f16x8 dequant_word(uint32_t w, const SZ& c) {
  const DqConst k = {kLoMask, kHiMask, kMagic, f16x2{(f16)0.0625f, (f16)0.0625f}};
  return dequant_word(w, c, k);
}

// ---------------------------------------------------------------------------
// bf16 activations (no counterpart in the reference, whose AWQ path is fp16 only: awq.py:42-43)
// ---------------------------------------------------------------------------
// Same packed weights, fp16 scales as the checkpoint holds them.  The weight of the bf16 path is
//   w = bf16((q - z) * s)    rounded once, to nearest even:
// q - z is formed exactly in fp16 as in the fp16 path (magic number, packed add / fma), the product with the scale
// exactly in fp32 (5 bits x 11 bits; v_fma_mix_f32 reads the fp16 half in place), one v_cvt_pk_bf16_f32 per pair.
// gfx950 has no packed bf16 arithmetic: 21 vector instructions per packed word against 13 for fp16.
typedef __bf16 bf16;
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct SZB {
  float s;    // the scale
  f16x2 nz1;  // as SZ
  f16x2 nz2;
};
__device__ __forceinline__ SZB expand_szb(uint32_t w) {
  const f16x2 v = as_h2(w);
  SZB r;
  r.s = (float)v.x;
  r.nz1 = f16x2{v.y, v.y};
  r.nz2 = r.nz1 + f16x2{(f16)960.0f, (f16)960.0f};
  return r;
}
// {bf16(e.x * s), bf16(e.y * s)}; the -0 addend keeps the sign of a zero product
__device__ __forceinline__ uint32_t scale_to_bf16x2(f16x2 e, float s) {
  f32x2 p;
  asm("v_fma_mix_f32 %0, %1, %2, neg(0) op_sel_hi:[1,0,0]" : "=v"(p.x) : "v"(e), "v"(s));
  asm("v_fma_mix_f32 %0, %1, %2, neg(0) op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(p.y) : "v"(e), "v"(s));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(p, bf16x2));
}
__device__ __forceinline__ bf16x8 dequant_word_bf16(uint32_t w, const SZB& c, const DqConst& k) {
  const uint32_t t = w >> 8;
  const f16x2 h0 = as_h2((w & k.lo) | k.magic);
  const f16x2 h1 = as_h2((w & k.hi) | k.magic);
  const f16x2 h2 = as_h2((t & k.lo) | k.magic);
  const f16x2 h3 = as_h2((t & k.hi) | k.magic);
  const f16x2 e0 = h0 + c.nz1;
  const f16x2 e1 = h1 * k.sixteenth + c.nz2;
  const f16x2 e2 = h2 + c.nz1;
  const f16x2 e3 = h3 * k.sixteenth + c.nz2;
  const u32x4 r = {scale_to_bf16x2(e0, c.s), scale_to_bf16x2(e1, c.s), scale_to_bf16x2(e2, c.s), scale_to_bf16x2(e3, c.s)};
  return __builtin_bit_cast(bf16x8, r);
}

// What a GEMM kernel needs to know about its activation / output type.
struct ActF16 {
  typedef f16 elem;
  typedef f16x4 x4;
  typedef f16x8 x8;
  typedef SZ consts;
  static __device__ __forceinline__ consts expand(uint32_t w) { return expand_sz(w); }
  static __device__ __forceinline__ x8 dequant(uint32_t w, const consts& c, const DqConst& k) { return dequant_word(w, c, k); }
  static __device__ __forceinline__ f32x4 mfma16(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x16 mfma32(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ x4 round(f32x4 v) { return x4{(f16)v.x, (f16)v.y, (f16)v.z, (f16)v.w}; }
  static __device__ __forceinline__ x4 add(x4 a, x4 b) { return a + b; }
};
struct ActBf16 {
  typedef bf16 elem;
  typedef bf16x4 x4;
  typedef bf16x8 x8;
  typedef SZB consts;
  static __device__ __forceinline__ consts expand(uint32_t w) { return expand_szb(w); }
  static __device__ __forceinline__ x8 dequant(uint32_t w, const consts& c, const DqConst& k) { return dequant_word_bf16(w, c, k); }
  static __device__ __forceinline__ f32x4 mfma16(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x16 mfma32(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ x4 round(f32x4 v) { return __builtin_convertvector(v, x4); }
  // bf16 + bf16: the exact sum rounded once (fp32 holds it unless the exponents are more than 16 apart)
  static __device__ __forceinline__ x4 add(x4 a, x4 b) {
    return __builtin_convertvector(__builtin_convertvector(a, f32x4) + __builtin_convertvector(b, f32x4), x4);
  }
};

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

// The same on bf16 results: silu in fp32 on the bf16 value, rounded to bf16, then a bf16 multiply (rounded once).
__device__ __forceinline__ bf16x4 silu_mul(bf16x4 g, bf16x4 u) {
  bf16x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x = (float)g[i];
    const bf16 sg = (bf16)(x / (1.0f + expf(-x)));
    r[i] = (bf16)((float)sg * (float)u[i]);
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
// Diagnostic build of ONE kernel translation unit (-DW4A16_STAMPS on that file alone, the others as shipped): that
// file's stamp buffer is read through this accessor, which each kernel file expands at its end.  (With every file
// built as one unit, -DW4A16_UNITY on w4a16.hip, the accessor lives there.)
#if defined(W4A16_STAMPS) && !defined(W4A16_UNITY)
#define W4A16_STAMPS_ACCESSOR                                                                                          \
  extern "C" int w4a16_dbg_stamps(void* host_out, size_t bytes) {                                                      \
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(w4a16::g_stamps), bytes, 0, hipMemcpyDeviceToHost);           \
  }
#else
#define W4A16_STAMPS_ACCESSOR
#endif
// LDS-DMA instructions that share an LDS base (M0) differ in their immediate offset (up to 4095), which moves the
// LDS address and the source address alike; see the activations' buffer descriptor in the 32x32x16 kernels.
constexpr uint32_t kDmaImmBias = 3072;
// A compile-time switch passed as a lambda argument.
template <int V>
struct Flag {
  static constexpr int value = V;
};

// Waits until at most N of the wave's vector-memory operations (which complete in issue order) are outstanding.
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
#ifndef W4A16_WAIT_ASM
  // s_waitcnt immediate of gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14 (the other counters at
  // their maximum: not waited for).  As a builtin the compiler's own wait-count bookkeeping sees it.
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
#else
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}

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

// Checking build of the 32x32x16 kernels (-DW4A16_CHECK_DMA, libw4a16_check.so; never the shipped library).  Their
// activation slabs arrive by LDS-DMA, whose completion the kernels await with hand-counted `s_waitcnt vmcnt(n)`: a
// miscount (an edit to the issue order, a compiler that reorders two loads) shows as a rare wrong tile.  Here every
// piece of LDS is filled with a poison pattern right before the DMA into it is issued, and every fragment read back
// is checked for the pattern: a read that overtakes its DMA is counted, every time.  g_dma_state[1] != 0 is the
// negative control of the test: the kernels then skip the counted waits AND read one piece back right behind the
// DMA instruction that fills it (the waits have slack whenever a slab arrives early, so skipping them alone need
// not show; a read a few cycles behind the DMA's issue cannot see its data).
#ifdef W4A16_CHECK_DMA
constexpr uint32_t kDmaPoison = 0xffffffffu;  // two fp16 NaNs no finite activation pair equals
#define W4A16_DMA_STATE_DECL static __device__ unsigned g_dma_state[2];
__device__ __forceinline__ void dma_check(const u32x4 v, unsigned* state) {
  if (v.x == kDmaPoison || v.y == kDmaPoison || v.z == kDmaPoison || v.w == kDmaPoison) atomicAdd(state, 1u);
}
#define W4A16_DMA_WAIT(N)                                   \
  do {                                                      \
    if (!dma_skip_waits) wait_vmcnt<(N)>();                 \
  } while (0)
#else
#define W4A16_DMA_WAIT(N) wait_vmcnt<(N)>()
#endif

// K split over workgroups of the private-slab 32x32x16 kernel (gridDim.y splits; one split: all null / whole K).
struct WideSplit {
  float* ws;          // partial sums: [output tile][split][wave][slice][64 lanes] x 16 bytes
  unsigned* tickets;  // arrival counters, one per output tile, zero between launches
  int ws_bytes;       // bytes behind `ws`
  int kb_per_split;   // 128-k blocks per workgroup
};

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
// host side
// ---------------------------------------------------------------------------
inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// The packed layout stores a multiple of 4 blocks of 128 k per column tile; the padding holds
// zero-valued weights (q = z), so kernels whose K-groups take blocks in lockstep never branch.
inline int padded_kblocks(int K) { return ceil_div(K, 512) * 4; }

inline int check_common(int K, int N, int G) {
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

inline int launched() {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    w4a16_note_hip_error((int)e);
    return W4A16_E_LAUNCH;
  }
  return W4A16_OK;
}

// Validates the arguments of the packed-layout GEMM entry points and fills the kernel argument block.
inline int fill_args(GemmArgs* a, const void* x, const void* packed_w, const void* packed_sz, const void* bias,
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

// j / m_blocks by multiply and shift for every work item j of a one-panel launch (exact while j * m_blocks < 65536)
// Kernels whose workgroups own a [rows] x [columns] tile list their work panel by panel: all row blocks of a slice next
// to each other up to 8 of them, panels of this many beyond (map_workgroup).
constexpr int kPanelRows = 4;
// The 32x32x16 kernels serve single problems with more rows than this (and K >= 2048).
constexpr int kWideFromRows = 16;

inline int row_block_magic(long long items, int m_blocks, int panel_rows) {
  if (panel_rows != m_blocks || items * m_blocks >= 65536) return 0;
  return (65536 + m_blocks - 1) / m_blocks;
}

// The launch shapes each kernel family is built for (the dispatchers and w4a16_built_shape expand these).
// (mt, t, waves) of the streaming kernel.  One or two 16-row tiles (more rows go to the other kernels); t in 1..8;
// 16 waves where 128 VGPRs per lane are enough, 4 / 8 for short K and for small workgroups, several to a CU.
#define W4A16_FOR_EACH_CFG(X) \
  X(1, 1, 16) X(1, 2, 16) X(1, 3, 16)                                     \
  X(1, 1, 8)  X(1, 2, 8)  X(1, 3, 8)  X(1, 4, 8)  X(1, 6, 8)  X(1, 8, 8)  \
  X(1, 1, 4)  X(1, 2, 4)  X(1, 3, 4)  X(1, 4, 4)                          \
  X(2, 1, 16)                                                             \
  X(2, 1, 8)  X(2, 2, 8)  X(2, 3, 8)  X(2, 4, 8)  X(2, 6, 8)              \
  X(2, 1, 4)  X(2, 2, 4)  X(2, 3, 4)
// (mt, t, waves) of the streaming kernel built for bf16 activations
#define W4A16_FOR_EACH_BF16(X) X(1, 1, 16) X(1, 2, 16) X(1, 3, 16) X(1, 2, 8) X(1, 4, 8) X(2, 1, 8) X(2, 2, 8) X(2, 3, 8) X(2, 4, 8)
// (mt, t, nw, kw) of the LDS-staged kernel (the shapes choose_cfg can return; the kernel serves what the 32x32x16
// kernels do not take: K not a multiple of 128, 24-32 rows of short-K or batched problems)
#define W4A16_FOR_EACH_TILED(X) \
  X(4, 3, 2, 4) X(4, 3, 4, 2) X(4, 3, 1, 4) X(4, 2, 2, 4) X(2, 3, 2, 4) X(2, 3, 1, 4) X(2, 3, 1, 8)
// (mt, t, kw) of the 32x32x16 kernel with private slabs: mt, t in 32-row / 32-column tiles
#ifndef W4A16_FOR_EACH_WIDE  // (kernel experiments build a single shape: -D'W4A16_FOR_EACH_WIDE(X)=X(2,3,4)')
#define W4A16_FOR_EACH_WIDE(X) \
  X(2, 3, 8) X(2, 2, 8) X(2, 1, 8) X(1, 3, 8) X(1, 2, 8) X(2, 3, 4) X(2, 2, 4) X(1, 3, 4) X(4, 3, 4) X(1, 1, 8)
#endif
// Slices (1 KiB: 4 accumulator registers x 64 lanes) per owner wave in the private-slab kernel's final exchange, and
// the workspace a K split over workgroups needs: the ticket region, then 4 owners x that many KiB per output tile
// and split.
// Private-slab shapes whose waves keep two whole activation slabs (the next block's arrives piece by piece during
// the current one): 4 waves x two row tiles, 128 KiB of LDS.
#ifndef W4A16_WIDE_DBL
#define W4A16_WIDE_DBL 1
#endif
constexpr bool wide_double_slab(int mt, int kw) { return W4A16_WIDE_DBL && mt == 2 && kw == 4; }
// private-slab shapes also built for bf16 activations (17 rows and more)
constexpr bool wide_bf16_built(int mt, int t, int kw) {
  return (mt == 2 && t == 3 && kw == 4) || (mt == 1 && t == 2 && kw == 8) || (mt == 4 && t == 3 && kw == 4) ||
         (mt == 1 && t == 3 && kw == 4) || (mt == 2 && t == 2 && kw == 4) || (mt == 1 && t == 1 && kw == 8);
}
constexpr int wide_slices_per_owner(int mt, int t, bool act) { return act ? ((t * mt + 1) / 2) * 2 : t * mt; }
inline size_t wide_ws_bytes(long long tiles, int splits, int mt, int t, bool act);
// shared-slab shapes also built for bf16 activations (a few hundred rows and more)
constexpr bool wide_shared_bf16_built(int mt, int t, int nw) {
  return mt == 4 && ((t == 2 && nw == 4) || (t == 3 && nw == 4) || (t == 2 && nw == 8));
}
// (mt, t, nw) of the shared-slab 32x32x16 kernel
#define W4A16_FOR_EACH_WIDE_SHARED(X) X(4, 1, 8) X(4, 2, 8) X(2, 1, 8) X(2, 2, 8) X(2, 3, 8) X(4, 3, 4) X(4, 2, 4)
// (t, waves) of the decode kernel; the fused activation takes the even t
#define W4A16_FOR_EACH_DECODE(X) \
  X(1, 16) X(2, 16) X(3, 16) X(1, 8) X(2, 8) X(3, 8) X(4, 8) X(6, 8) X(8, 8) X(2, 4) X(4, 4)

// ---- decode kernel (up to 4 rows) ----
constexpr int kDecodeMaxRows = 4;    // with LDS staging, the compact reduction and the optional K split
constexpr int kDecodeMaxRowsL2 = 16;  // one 16-row tile: through the LDS ring or with fragments straight from L2 (no K split)
constexpr int kDecodeMaxRowsTwo = 32;  // two 16-row tiles through the LDS ring (plain GEMM, one block in flight)
// Two row tiles: each (tile, row tile) is summed by a wave of its own; a wave parks its partial tiles in its own 8-KiB
// ring slot where they fit (up to 4 column tiles), else behind the ring.
constexpr size_t decode_two_row_tiles_lds(int t, int waves, int db) {
  return (size_t)waves * db * 8192 + (2 * t * 1024 <= db * 8192 ? 0 : (size_t)waves * 2 * t * 1024);
}
// Built where it is ahead of the 32x32x16 kernel with weights coming from HBM: one column tile x 16 waves, i.e. layers
// of up to 256 tiles (4096 columns), which it covers with twice the workgroups of the 32-column shape <1,1,8>
// (4096 x 4096 at 17 / 32 rows 6.15 / 7.2 us against 6.8 / 7.7, 11008 x 4096 12.2 / 15.4 against 13.8 / 17.9); with two
// tiles it draws level with <1,1,8>, with three x 16 waves it runs out of registers (128 per lane).
constexpr bool decode_two_row_tiles_built(int t, int waves, int db) {
  return db == 1 && waves == 16 && t == 1 && decode_two_row_tiles_lds(t, waves, db) <= 160 * 1024;
}
constexpr int kMaxSplits = 8;

// What the decode kernel's addressing admits: whole 128-k blocks, 32-bit byte offsets, and a group size whose
// group index is a shift or one multiply away from the block index.
inline bool decode_ok(const GemmArgs& a) {
  if (a.M > kDecodeMaxRowsTwo || a.K % 128 || (long long)a.M * a.K * 2 >= (1LL << 31)) return false;
  if (!(a.G == 32 || a.G == 64 || (a.G % 128 == 0 && a.G / 128 <= 64))) return false;
  if (a.KB >= 1024) return false;
  if ((long long)a.NT * a.KBS * 1024 >= (1LL << 32) || (long long)a.NT * a.NG * 64 >= (1LL << 31)) return false;
  return true;
}

// Bytes of workspace a K-split launch needs: one slot of fp32 partial sums per (problem, workgroup column,
// split), then one ticket counter per (problem, workgroup column).
inline size_t decode_partial_bytes(int batch, int gx, int splits, bool act) {
  return (size_t)batch * gx * splits * (act ? 2048 : 1024);
}
// The ticket counters come first, in a region of fixed size: wherever a launch's partial sums land, they never
// overwrite a counter that another launch shape expects to find at zero.
constexpr int kMaxTickets = 16384;
constexpr size_t kTicketBytes = (size_t)kMaxTickets * 4;
inline size_t decode_ws_bytes(int batch, int gx, int splits, bool act) {
  return splits <= 1 ? 0 : kTicketBytes + decode_partial_bytes(batch, gx, splits, act);
}

// scale/zero words per 128-k block and tile, and the LDS a workgroup needs: per wave 1 KiB of activations per
// block, then the reduction's partial sums
inline int decode_nsz(int G) { return G % 128 == 0 ? 1 : G == 64 ? 2 : 4; }
inline size_t decode_lds_bytes(int t, int waves, int nb_max, int M, int G) {
  (void)G;
  if (M > kDecodeMaxRows) return (size_t)waves * t * 1024;  // whole partial tiles, no staged activations
  return (size_t)waves * nb_max * 1024 + (size_t)waves * t * 256;
}


inline size_t wide_ws_bytes(long long tiles, int splits, int mt, int t, bool act) {
  return splits <= 1 ? 0 : kTicketBytes + (size_t)tiles * splits * 4 * wide_slices_per_owner(mt, t, act) * 1024;
}

inline bool decode_built(int t, int waves, bool act) {
  if (act && t % 2 != 0) return false;
#define X(T_, W_) \
  if (t == T_ && waves == W_) return true;
  W4A16_FOR_EACH_DECODE(X)
#undef X
  return false;
}

// The 8-wave shapes with two 32-row tiles run the two waves of a SIMD half a block apart (see the kernel): the
// conversion is half of such a wave's issue slots, and a wave that converts while its partner is at the barrier keeps
// the matrix pipe fed (measured on 4096 x 11008: 2 to 7 % at 512 to 4096 rows).  With four row tiles the matrix
// instructions dominate either way and the stagger measured 0 to -1 %: those keep the plain loop.
#ifndef W4A16_STAGGER
#define W4A16_STAGGER 1
#endif
template <int MT, int T, int NW>
constexpr bool wide_shared_staggered() { return W4A16_STAGGER && NW == 8 && MT == 2; }
// Slabs in the ring of gemm_wide_shared_kernel: three (the DMA runs two blocks ahead) for workgroups that have a CU to
// themselves; two for the 4-wave shape that leaves registers and LDS for a second workgroup on the CU, whose
// arithmetic then covers the first one's waits, prologue and stores; four for the staggered shapes, where a slab is
// read for two barrier intervals.
template <int MT, int T, int NW>
constexpr int ring_slabs() { return wide_shared_staggered<MT, T, NW>() ? 4 : (NW == 4 && MT * T <= 8) ? 2 : 3; }

// One dispatcher per kernel family, each defined in that family's translation unit: launch shape -> instantiation.
W4A16_HIDDEN int dispatch_packed(const GemmArgs& a, int mt, int t, int waves, hipStream_t s, int batch = 1,
                                 bool act = false);
// bf16 activations: the streaming kernel's structure at any row count (row blocks of 16 or 32 in gridDim.y)
W4A16_HIDDEN int dispatch_packed_bf16(const GemmArgs& a, int mt, int t, int waves, hipStream_t s);
W4A16_HIDDEN int dispatch_tiled(const GemmArgs& a, int mt, int t, int nw, int kw, hipStream_t s, int batch = 1,
                                bool act = false);
// `splits` > 1: K split over workgroups through the workspace `ws` (4-wave shapes)
// `bf16`: bf16 activations (the shapes of wide_bf16_built, one problem, no fused activation, no split)
W4A16_HIDDEN int dispatch_wide(const GemmArgs& a, int mt, int t, int kw, hipStream_t s, int batch = 1,
                               bool act = false, int splits = 1, void* ws = nullptr, size_t ws_bytes = 0,
                               bool bf16 = false);
W4A16_HIDDEN int dispatch_wide_shared(const GemmArgs& a, int mt, int t, int nw, hipStream_t s, int batch = 1,
                                      bool act = false, bool bf16 = false);
// `depth`: blocks a wave keeps in flight (1 or 2; 0 = the default for the shape)
W4A16_HIDDEN int dispatch_decode(const GemmArgs& a, int t, int waves, hipStream_t s, int batch, bool act, int splits,
                                 void* ws, size_t ws_bytes, int depth = 0, bool bf16 = false);

}  // namespace w4a16
