// libw4a16.so: what runs on or produces the tensors themselves: the GEMM on the stock AWQ layout, dequantisation
// (stock and private layout), the one-time re-layout, and the streaming-read measurement aid; kernels and their C
// entry points.  Layout, arithmetic and the kernel families are described at the top of w4a16.hip.

#include "w4a16_common.h"

namespace W4A16_HIDDEN w4a16 {
namespace {

// Logical element i of a packed word sits in nibble kOrder[i].
static __device__ __constant__ const int kOrderDev[8] = {0, 4, 1, 5, 2, 6, 3, 7};

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

}  // namespace
}  // namespace w4a16

using namespace w4a16;

extern "C" {

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
        w4a16_note_hip_error((int)e);                                                                         \
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
