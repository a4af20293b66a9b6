// libw4a16.so: the streaming GEMM kernel (v_mfma_f32_16x16x32_f16, up to 32 rows) and its launcher.
// Layout, arithmetic and the kernel families are described at the top of w4a16.hip.

#include "w4a16_common.h"

namespace W4A16_HIDDEN w4a16 {
namespace {

// MT : 16-row tiles of the output per workgroup
// T  : 16-column tiles per workgroup
// W  : wavefronts per workgroup (they split K)
// SZS: k-steps (of 32) covered by one scale/zero word: 4 when G % 128 == 0,
//      2 when G == 64, 1 for any other multiple of 32
// ACT: the weight is [gate | up] (two halves of N/2 columns, N/2 a multiple of 16); the workgroup
//      takes T/2 column tiles of each half and stores silu(gate) * up, an [M, N/2] output.
// AT : the activation / output type (ActF16; ActBf16 for the bf16 entry point)
template <int MT, int T, int W, int SZS, bool ACT = false, class AT = ActF16>
__global__ void __launch_bounds__(64 * W) gemm_packed_kernel(const GemmArgs ga) {
  typedef typename AT::elem elem;
  typedef typename AT::x4 ex4;
  typedef typename AT::x8 ex8;
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
  const elem* xp[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = min(m0 + mt * 16 + n16, a.M - 1);
    xp[mt] = reinterpret_cast<const elem*>(a.x) + (size_t)row * a.K + kq * 8;
  }

  f32x4 acc[T][MT];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const DqConst dqc = make_dq_const();

  // The x fragments of a whole block (4 k-steps) travel with the block's weights: every load of a block is
  // issued up front, oldest-needed first (x from L2, the scale/zero words, then the weight stream from HBM),
  // and nothing is loaded inside the compute phase.  (16 * MT registers per buffer: the kernel is built for one
  // and two row tiles; more rows go to the kernels that stage activations in LDS.)
  static_assert(MT <= 2, "x fragments of a whole block are held in registers");
  struct Block {
    ex8 x[4][MT];
    uint32_t sz[T][NSZ];
    u32x4 q[T];
  };

  // k-steps past K (K not a multiple of 128) re-read k = 0: the packer fills k >= K with q = z, i.e. exact
  // zero weights.
  auto load_x = [&](int k, ex8 (&xf)[MT]) {
    const int ke = (k < a.K) ? k : 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xf[mt] = *reinterpret_cast<const ex8*>(xp[mt] + ke);
  };

  auto load_block = [&](int kb, Block& b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) load_x(kb * 128 + j * 32, b.x[j]);
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

  // One 128-k block: 4 k-steps of (T dequantised A fragments) x (MT x fragments), tile by tile, in the order the
  // weight loads were issued: the wave starts on its first tile while the others are still on their way (a
  // wave's loads complete in issue order, so the compiler's wait before tile t lets the loads of tiles t + 1..
  // stay outstanding).
  auto compute_block = [&](const Block& b) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      typename AT::consts ct;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j % SZS == 0) ct = AT::expand(b.sz[t][j / SZS]);
        const ex8 af = (W4A16_DBG & 8) ? __builtin_bit_cast(ex8, b.q[t]) : AT::dequant(b.q[t][j], ct, dqc);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[t][mt] = AT::mfma16(af, b.x[j][mt], acc[t][mt]);
      }
    }
  };

  // K loop.  Blocks past KB are never computed.  One-tile workgroups (one MFMA per weight fragment) keep three
  // blocks in flight ahead of the one being computed instead of one: a wave's block is only 1 KiB of weights,
  // and with a long K (11008) one block ahead leaves the CU with too few bytes in flight to cover HBM latency.
  if constexpr (T * MT == 1) {
    Block b0, b1, b2, b3;
    int kb = w;
    if (kb < a.KB) load_block(kb, b0);
    if (kb + W < a.KB) load_block(kb + W, b1);
    if (kb + 2 * W < a.KB) load_block(kb + 2 * W, b2);
    while (kb < a.KB) {
      if (kb + 3 * W < a.KB) load_block(kb + 3 * W, b3);
      compute_block(b0);
      kb += W;
      if (kb >= a.KB) break;
      if (kb + 3 * W < a.KB) load_block(kb + 3 * W, b0);
      compute_block(b1);
      kb += W;
      if (kb >= a.KB) break;
      if (kb + 3 * W < a.KB) load_block(kb + 3 * W, b1);
      compute_block(b2);
      kb += W;
      if (kb >= a.KB) break;
      if (kb + 3 * W < a.KB) load_block(kb + 3 * W, b2);
      compute_block(b3);
      kb += W;
    }
  } else {
    Block b0, b1;
    int kb = w;
    if (kb < a.KB) load_block(kb, b0);
    STAMP(1);
    while (kb < a.KB) {
      int kn = kb + W;
      if (kn < a.KB) load_block(kn, b1);
      compute_block(b0);
#ifdef W4A16_STAMPS
      STAMP(2 + min(n_done, 7)); ++n_done;
#endif
      kb = kn;
      if (kb >= a.KB) break;
      kn = kb + W;
      if (kn < a.KB) load_block(kn, b0);
      compute_block(b1);
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
    ex4 h = AT::round(v);
    if (a.bias != nullptr) h = AT::add(h, *reinterpret_cast<const ex4*>(reinterpret_cast<const elem*>(a.bias) + n_bias));
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
        *reinterpret_cast<ex4*>(reinterpret_cast<elem*>(a.out) + (size_t)m * NH + n) = silu_mul(finish(v, n), finish(vu, NH + n));
    } else {
      if (nt < a.NT && n < a.N && m < a.M) {
        ex4* dst = reinterpret_cast<ex4*>(reinterpret_cast<elem*>(a.out) + (size_t)m * a.N + n);
        ex4 h = finish(v, n);
        // fp16 + fp16, as `residual + linear(x)` on fp16 tensors; one lane owns each element
        if (a.add_out) h = AT::add(*dst, h);
        *dst = h;
      }
    }
  };
  constexpr int OUTS = H * MT;  // stores per workgroup: tiles, or gate/up tile pairs
  if (W > 1 && MT == 1 && a.M <= 4) {
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
        if (nt < NTH) *reinterpret_cast<ex4*>(reinterpret_cast<elem*>(a.out) + (size_t)m * NH + n) = silu_mul(finish(v, n), finish(vu, NH + n));
      } else {
        if (nt < a.NT && n < a.N) {
          ex4* dst = reinterpret_cast<ex4*>(reinterpret_cast<elem*>(a.out) + (size_t)m * a.N + n);
          ex4 hv = finish(v, n);
          if (a.add_out) hv = AT::add(*dst, hv);
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
      w4a16_note_hip_error((int)e);
      return W4A16_E_LAUNCH;
    }
  }
  hipLaunchKernelGGL(kern, grid, block, lds, s, a);
  return launched();
}

// bf16 activations: plain GEMM (+ bias, + add-to-out), one problem per launch
template <int MT, int T, int W>
int launch_packed_bf16(const GemmArgs& a, hipStream_t s) {
  const size_t lds = (size_t)reduce_lds_tiles(W, T * MT) * 1024;
  const dim3 grid(ceil_div(a.NT, T), ceil_div(a.M, MT * 16), 1);
  void (*kern)(const GemmArgs) = nullptr;
  if (a.G % 128 == 0)
    kern = gemm_packed_kernel<MT, T, W, 4, false, ActBf16>;
  else if (a.G == 64)
    kern = gemm_packed_kernel<MT, T, W, 2, false, ActBf16>;
  else
    kern = gemm_packed_kernel<MT, T, W, 1, false, ActBf16>;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      w4a16_note_hip_error((int)e);
      return W4A16_E_LAUNCH;
    }
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * W), lds, s, a);
  return launched();
}

}  // namespace

int dispatch_packed_bf16(const GemmArgs& a, int mt, int t, int waves, hipStream_t s) {
#define X(MT_, T_, W_) \
  if (mt == MT_ && t == T_ && waves == W_) return launch_packed_bf16<MT_, T_, W_>(a, s);
  W4A16_FOR_EACH_BF16(X)
#undef X
  return W4A16_E_SHAPE;
}

int dispatch_packed(const GemmArgs& a, int mt, int t, int waves, hipStream_t s, int batch,
                    bool act) {
#define X(MT_, T_, W_) \
  if (mt == MT_ && t == T_ && waves == W_) return launch_packed<MT_, T_, W_>(a, s, batch, act);
  W4A16_FOR_EACH_CFG(X)
#undef X
  return W4A16_E_SHAPE;
}


}  // namespace w4a16

W4A16_STAMPS_ACCESSOR
