// libw4a16.so: the 32x32x16 GEMM kernel with one activation slab per workgroup (many rows) and its launcher.
// Layout, arithmetic and the kernel families are described at the top of w4a16.hip.

#include "w4a16_common.h"

#ifdef W4A16_CHECK_DMA
namespace W4A16_HIDDEN w4a16 {
W4A16_DMA_STATE_DECL
}
#endif

namespace W4A16_HIDDEN w4a16 {
namespace {

// ---------------------------------------------------------------------------
// 32x32x16 GEMM for many rows: one activation slab shared by all waves of the workgroup
// ---------------------------------------------------------------------------
// With hundreds of rows and more the workgroup's traffic, not its arithmetic, decides: a [32 MT] x [32 T NW]
// tile re-reads its activation rows once per workgroup, so the wider the tile the fewer bytes per
// matrix instruction.  Here the NW waves of a workgroup each own T 32-column tiles for all MT 32-row
// tiles over the WHOLE of K (no K split, so no cross-wave reduction: results go from the accumulators
// straight to memory), and share the [32 MT rows] x [128 k] activation slab of the current block, which
// the waves fetch together by LDS-DMA into a ring of slabs (two to four, see ring_slabs(); 256-byte rows, 16-byte
// chunks XOR-swizzled by the row), two blocks ahead.  A weight fragment converted once feeds MT matrix
// instructions: with MT = 4 the conversion takes a quarter of the vector issue slots of the 16x16x32
// kernels'.  One workgroup barrier per 128-k block: after it every wave's share of the block's slab has
// landed (each waits for its own DMA before arriving) and the slab of the block before last is free.

// AT: the activation / output type (fp16, or bf16 for the plain GEMM of a few shapes; see dequant_word_bf16).
template <int MT, int T, int NW, int SZS, bool ACT = false, class AT = ActF16>
__global__ void __launch_bounds__(64 * NW, (ring_slabs<MT, T, NW>() == 2 ? 2 : 1)) gemm_wide_shared_kernel(int n_slices, int m_blocks, int panel_rows,
                                                                   const GemmArgs ga) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub
  typedef typename AT::elem elem;
  typedef typename AT::x4 ex4;
  typedef typename AT::x8 ex8;
  typedef typename AT::consts SZc;
  static_assert(!ACT || std::is_same<AT, ActF16>::value, "the fused activation is built for fp16");
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
  constexpr bool STG = wide_shared_staggered<MT, T, NW>();  // the waves of a SIMD run half a block apart
#ifndef W4A16_SHARED_SPREAD
#define W4A16_SHARED_SPREAD 1
#endif
  constexpr bool SPREAD = W4A16_SHARED_SPREAD && !STG && RING == 3;  // requests dealt out between the matrix instructions
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r32 = lane & 31, h = lane >> 5;
#ifdef W4A16_CHECK_DMA
  const bool dma_skip_waits = __builtin_amdgcn_readfirstlane(g_dma_state[1]) != 0;
#endif

#ifdef W4A16_STAMPS
  constexpr int NWAVES = NW;
  int n_done = 0;
  if (lane == 0 && STAMP_OK) g_stamps[((size_t)blockIdx.x * NWAVES + w) * 32 + 30] = __builtin_amdgcn_s_memrealtime();
#endif
  STAMP(0);
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
  // this wave's pieces of a slab: piece p = ND w + j covers rows 4 p .. 4 p + 3; lane -> row 4 p + (lane >> 4),
  // 16-byte position lane & 15, which holds chunk (lane & 15) ^ (row & 15) of the row's 128 k.  A wave's pieces are
  // neighbours in LDS: up to four of them share one LDS base (M0) and differ in the instruction's immediate offset,
  // which moves the LDS address and the source address alike, so the lane offsets are smaller by the same amount.
  uint32_t xoff[ND];
#pragma unroll
  for (int j = 0; j < ND; ++j) {
    const int row = 4 * (w * ND + j) + (lane >> 4);
    xoff[j] = ((uint32_t)min(m0 + row, a.M - 1) * a.K + ((lane & 15) ^ (row & 15)) * 8) * 2u + kDmaImmBias - (uint32_t)(j & 3) * 1024u;
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
  // (the activations' descriptor starts kDmaImmBias bytes BEFORE the matrix and every lane offset is larger by that
  // much: a DMA instruction with an immediate offset then takes it off its lane offset without going below zero)
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<unsigned char*>(const_cast<f16*>(a.x)) - kDmaImmBias, 0, (int)((uint32_t)a.M * (uint32_t)a.K * 2u + kDmaImmBias), 0x00020000);
  struct WSet {
    uint32_t sz[T][NSZ];
    u32x4 q[T][2];
  };
  auto load_w = [&](int kb, WSet& b) {
    int goff[NSZ];  // (once per block, not per tile: the waves of a workgroup share one scalar unit)
#pragma unroll
    for (int s = 0; s < NSZ; ++s) goff[s] = min(group_of<SZS>(a, kb * 128 + s * SZS * 32), a.NG - 1) * 64;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int s = 0; s < NSZ; ++s) b.sz[t][s] = __builtin_amdgcn_raw_buffer_load_b32(sz_rs, szoff[t], goff[s], 0);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      b.q[t][0] = __builtin_amdgcn_raw_buffer_load_b128(w_rs, woff[t], kb * 1024, 0);
      b.q[t][1] = __builtin_amdgcn_raw_buffer_load_b128(w_rs, woff[t] + 512, kb * 1024, 0);
    }
  };
  auto dma_x = [&](int kb, int buf) {
#ifdef W4A16_CHECK_DMA
#pragma unroll
    for (int j = 0; j < ND; ++j)
      *reinterpret_cast<u32x4*>(smem_raw + buf * SLAB + (w * ND + j) * 1024 + lane * 16) = u32x4{kDmaPoison, kDmaPoison, kDmaPoison, kDmaPoison};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the pattern is in place before the DMA can land on it
#endif
    auto issue = [&](int j, auto imm_) {  // (the immediate has to be a literal)
      constexpr int IMM = decltype(imm_)::value;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          x_rs, (__attribute__((address_space(3))) void*)(smem_raw + buf * SLAB + (w * ND + (j & ~3)) * 1024), 16,
          xoff[j], kb * 256, IMM, 0);
    };
#pragma unroll
    for (int j = 0; j < ND; ++j) {
      switch (j & 3) {
        case 0: issue(j, Flag<0>{}); break;
        case 1: issue(j, Flag<1024>{}); break;
        case 2: issue(j, Flag<2048>{}); break;
        default: issue(j, Flag<3072>{}); break;
      }
    }
#ifdef W4A16_CHECK_DMA
    if (dma_skip_waits) dma_check(*reinterpret_cast<const u32x4*>(smem_raw + buf * SLAB + w * ND * 1024 + lane * 16), &g_dma_state[0]);
#endif
  };
  // One of a wave's requests for the next blocks, by index: the WL register loads of weight block `kb_w` (scale/zero
  // words first), then its ND DMA pieces of block `kb_d`'s slab.  (Indices are constants once the callers' loops are
  // unrolled; `goff` as in load_w.)
  auto request = [&](auto lw_, auto dma_, int k, int kb_w, const int (&goff)[NSZ], WSet& b, int kb_d, int buf) {
    constexpr bool LW = decltype(lw_)::value, DMA = decltype(dma_)::value;
    if (k < T * NSZ) {
      if constexpr (LW) b.sz[k / NSZ][k % NSZ] = __builtin_amdgcn_raw_buffer_load_b32(sz_rs, szoff[k / NSZ], goff[k % NSZ], 0);
    } else if (k < WL) {
      const int t = (k - T * NSZ) >> 1, hf = (k - T * NSZ) & 1;
      if constexpr (LW) b.q[t][hf] = __builtin_amdgcn_raw_buffer_load_b128(w_rs, woff[t] + hf * 512, kb_w * 1024, 0);
    } else if (k < WL + ND) {
      if constexpr (DMA) {
        const int j = k - WL;
        auto issue = [&](auto imm_) {
          constexpr int IMM = decltype(imm_)::value;
#ifdef W4A16_CHECK_DMA
          *reinterpret_cast<u32x4*>(smem_raw + buf * SLAB + (w * ND + j) * 1024 + lane * 16) = u32x4{kDmaPoison, kDmaPoison, kDmaPoison, kDmaPoison};
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              x_rs, (__attribute__((address_space(3))) void*)(smem_raw + buf * SLAB + (w * ND + (j & ~3)) * 1024), 16,
              xoff[j], kb_d * 256, IMM, 0);
        };
        switch (j & 3) {
          case 0: issue(Flag<0>{}); break;
          case 1: issue(Flag<1024>{}); break;
          case 2: issue(Flag<2048>{}); break;
          default: issue(Flag<3072>{}); break;
        }
      }
    }
  };
  auto read_x = [&](int ks, int base, ex8 (&xf)[MT]) {
#ifdef W4A16_SH_DBG
    if ((W4A16_SH_DBG & 2) && ks > 0) return;  // ablation: one fragment read per block
#endif
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      xf[mt] = *reinterpret_cast<const ex8*>(smem_raw + base + (((2 * ks + h) ^ xsw) << 4) + mt * 8192);
#ifdef W4A16_CHECK_DMA
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      dma_check(__builtin_bit_cast(u32x4, xf[mt]), &g_dma_state[0]);
#endif
  };
  auto convert = [&](int kstep, int t, const WSet& b, SZc (&c)[T]) {
    if (kstep % (2 * SZS) == 0) c[t] = AT::expand(b.sz[t][kstep / (2 * SZS)]);
#ifdef W4A16_SH_DBG
    if (W4A16_SH_DBG & 1) return __builtin_bit_cast(ex8, b.q[t][kstep & 1]);  // ablation: no conversion
#endif
    return AT::dequant(b.q[t][kstep & 1][kstep >> 1], c[t], dqc);
  };
  auto math = [&](int kstep, const WSet& b, SZc (&c)[T], const ex8 (&xf)[MT], ex8& af0) {
    ex8 af[T + 1];
    af[0] = af0;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (t + 1 < T)
        af[t + 1] = convert(kstep, t + 1, b, c);
      else if (kstep + 1 < 8)
        af[T] = convert(kstep + 1, 0, b, c);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[t][mt] = AT::mfma32(af[t], xf[mt], acc[t][mt]);
    }
    if (kstep + 1 < 8) af0 = af[T];
    constexpr int per = (13 + MT - 1) / MT;
#pragma unroll
    for (int i = 0; i < T * MT; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, per, 0);
    }
  };
  // The same arithmetic tile by tile, each tile's matrix instructions and the next tile's conversion in a scheduling
  // region of their own, and `between(slot)` (slot = T kstep + t: a request for the next blocks) issued in program
  // order behind tile t's region, in the shadow of its last matrix instruction.
  auto math_spread = [&](int kstep, const WSet& b, SZc (&c)[T], const ex8 (&xf)[MT], ex8& af0, auto&& between) {
    ex8 af = af0;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      ex8 nx = af;
      if (t + 1 < T)
        nx = convert(kstep, t + 1, b, c);
      else if (kstep + 1 < 8)
        nx = convert(kstep + 1, 0, b, c);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[t][mt] = AT::mfma32(af, xf[mt], acc[t][mt]);
      constexpr int per = (13 + MT - 1) / MT;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x002, per, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      af = nx;
      __builtin_amdgcn_sched_barrier(0);
      between(kstep * T + t);
      __builtin_amdgcn_sched_barrier(0);
    }
    af0 = af;
  };
  // Four k-steps (half HB of a block) from weight set `b`.  On entry `xa` holds the fragments of the half's first
  // k-step and `af0` tile 0's converted weights for it; on exit `af0` holds tile 0's weights of the k-step that
  // follows the half (if the block has one): the conversion runs one tile ahead of the matrix instructions.
  auto half = [&](auto hb_, int base, const WSet& b, SZc (&c)[T], ex8& af0, ex8 (&xa)[MT]) {
    constexpr int K0 = 4 * decltype(hb_)::value;
    ex8 xb[MT];
#pragma unroll
    for (int k2 = 0; k2 < 4; k2 += 2) {
      read_x(K0 + k2 + 1, base, xb);
      __builtin_amdgcn_sched_barrier(0);
      math(K0 + k2, b, c, xa, af0);
      __builtin_amdgcn_sched_barrier(0);
      if (k2 == 0) read_x(K0 + 2, base, xa);
      __builtin_amdgcn_sched_barrier(0);
      math(K0 + k2 + 1, b, c, xb, af0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  const int n = a.KB;  // >= 2 (host-checked)
  if constexpr (!STG) {
  // Block i from weight set `cur` in ring slab i % 3.  A wave's vector-memory operations, in issue order:
  //   D(0), D(1), W(0) | D(2), W(1) | D(3), W(2) | ..      (D = its ND DMA loads of a slab, W = WL loads)
  // one group per block, issued right after the block's barrier.  Before the barrier of block i the wave's
  // own share of D(i) must have landed: all but the ND + WL youngest (D(i+1), W(i)) are done then; in the
  // last block D(i+1) does not exist: all but the WL youngest.
  // DMA 0/1: whether D(i+2) exists; LW: whether W(i+1) exists.
  auto block = [&](auto dma, auto lw, int i, const WSet& cur, WSet& nxt) {
    constexpr bool DMA = decltype(dma)::value, LW = decltype(lw)::value;
    const int base = (i % RING) * SLAB + xrd;
#ifdef W4A16_STAMPS
    if (n_done < 8) STAMP(2 + 3 * n_done);
#endif
    // (two slabs: D(i+1) is issued after this barrier, so only W(i) is younger than D(i))
    W4A16_DMA_WAIT((LW && RING == 3) ? ND + WL : WL);
#ifdef W4A16_STAMPS
    if (n_done < 8) STAMP(3 + 3 * n_done);
#endif
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#ifdef W4A16_STAMPS
    if (n_done < 8) STAMP(4 + 3 * n_done);
    ++n_done;
#endif
    if constexpr (SPREAD) {
      // The block's requests (W(i + 1), then this wave's pieces of D(i + 2)) go out one or two at a time between the
      // tiles' matrix instructions instead of in one burst behind the barrier: all waves of the workgroup pass the
      // barrier together, and the CU's address path takes their NW x (ND + WL) requests at ~16 cycles each while
      // every wave's in-order issue waits behind its own.  The waits above count a block's requests, not their order.
      int goff[NSZ];
#pragma unroll
      for (int s = 0; s < NSZ; ++s) goff[s] = min(group_of<SZS>(a, (i + 1) * 128 + s * SZS * 32), a.NG - 1) * 64;
      constexpr int OPS = WL + ND, SLOTS = 8 * T, PER = (OPS + SLOTS - 1) / SLOTS;
      auto between = [&](int slot) {
#pragma unroll
        for (int u = 0; u < PER; ++u)
          if (slot * PER + u < OPS) request(lw, dma, slot * PER + u, i + 1, goff, nxt, i + RING - 1, (i + RING - 1) % RING);
      };
      SZc c[T];
      ex8 xa[MT], xb[MT];
      read_x(0, base, xa);
      ex8 af0 = convert(0, 0, cur, c);
#pragma unroll
      for (int k2 = 0; k2 < 8; k2 += 2) {
        read_x(k2 + 1, base, xb);
        __builtin_amdgcn_sched_barrier(0);
        math_spread(k2, cur, c, xa, af0, between);
        if (k2 + 2 < 8) read_x(k2 + 2, base, xa);
        __builtin_amdgcn_sched_barrier(0);
        math_spread(k2 + 1, cur, c, xb, af0, between);
      }
      return;
    }
    if constexpr (DMA) dma_x(i + RING - 1, (i + RING - 1) % RING);
    __builtin_amdgcn_sched_barrier(0);  // (the waits above count a block's requests, whatever their order)
    if constexpr (LW) load_w(i + 1, nxt);
    __builtin_amdgcn_sched_barrier(0);
    SZc c[T];
    ex8 xa[MT], xb[MT];
    read_x(0, base, xa);
    ex8 af0 = convert(0, 0, cur, c);
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

  WSet w0, w1;
  dma_x(0, 0);
  if constexpr (RING == 3) dma_x(1, 1);
  __builtin_amdgcn_sched_barrier(0);
  load_w(0, w0);
  __builtin_amdgcn_sched_barrier(0);
  STAMP(1);
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

  } else {
    // Staggered: the two waves of a SIMD (w and w + NW / 2) run half a block apart, so that one is in the middle of
    // its matrix instructions while the other passes the barrier, requests its loads and converts its first
    // fragment.  Between barrier j and barrier j + 1 the waves of the first half (A) work through block j; those of
    // the second half (B) finish block j - 1 (its k-steps 4-7) and start block j (k-steps 0-3).  Block j's slab is
    // thus read until barrier j + 2, and the DMA issued behind barrier j (for block j + 2) goes to the slot of
    // block j - 2: a ring of four.  Every wave still fetches its share of every slab.  A wave's vector-memory
    // operations, in issue order:  D(0), D(1), W(0) | D(2), W(1) | D(3), W(2) | ..   (one group per barrier; B
    // issues W(j + 1) in the middle of the interval, once the set of block j - 1 is free).  Before barrier j the
    // wave's share of D(j) must have landed: all but the youngest ND + WL (D(j + 1), W(j)); WL where D(j + 1)
    // does not exist.
    static_assert(RING == 4 && NW % 2 == 0, "ring of four slabs");
    auto slab_of = [&](int j) { return (j & 3) * SLAB + xrd; };
    WSet w0, w1;
    dma_x(0, 0);
    dma_x(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    load_w(0, w0);
    __builtin_amdgcn_sched_barrier(0);
    if (w < NW / 2) {
      // ---- A: whole blocks, as the unstaggered loop ----
      auto block_a = [&](auto dma, auto lw, int j, const WSet& cur, WSet& nxt) {
        constexpr bool DMA = decltype(dma)::value, LW = decltype(lw)::value;
        W4A16_DMA_WAIT(LW ? ND + WL : WL);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DMA) dma_x(j + 2, (j + 2) & 3);
        __builtin_amdgcn_sched_barrier(0);  // the wait counts rely on this issue order: DMA first, then weights
        if constexpr (LW) load_w(j + 1, nxt);
        __builtin_amdgcn_sched_barrier(0);
        const int base = slab_of(j);
        SZc c[T];
        ex8 xa[MT];
        read_x(0, base, xa);
        ex8 af0 = convert(0, 0, cur, c);
        half(Flag<0>{}, base, cur, c, af0, xa);
        read_x(4, base, xa);
        half(Flag<1>{}, base, cur, c, af0, xa);
      };
      int j = 0;
      for (; j + 2 <= n - 2; j += 2) {
        block_a(Flag<1>{}, Flag<1>{}, j, w0, w1);
        block_a(Flag<1>{}, Flag<1>{}, j + 1, w1, w0);
      }
      if (j < n - 2) {
        block_a(Flag<1>{}, Flag<1>{}, j, w0, w1);
        w0 = w1;
        ++j;
      }
      block_a(Flag<0>{}, Flag<1>{}, j, w0, w1);
      block_a(Flag<0>{}, Flag<0>{}, j + 1, w1, w0);
    } else {
      // ---- B: half a block behind: the barriers fall in the middle of its blocks ----
      // Block b: W(b + 1) requested, k-steps 0-3, barrier b + 1 (behind it the DMA for block b + 3), k-steps 4-7.
      // BAR: barrier b + 1 and W(b + 1) exist (b + 1 < n); NXT: D(b + 2) exists; DMA: D(b + 3) exists.  Before barrier
      // b + 1 this wave's share of D(b + 1) has landed once all but the youngest ND + WL (D(b + 2), W(b + 1)) are done.
      auto block_b = [&](auto bar, auto nxt_, auto dma, int b, const WSet& cur, WSet& nxt) {
        constexpr bool BAR = decltype(bar)::value, NXT = decltype(nxt_)::value, DMA = decltype(dma)::value;
        constexpr bool LW = BAR;
        if constexpr (LW) load_w(b + 1, nxt);
        __builtin_amdgcn_sched_barrier(0);
        const int base = slab_of(b);
        SZc c[T];
        ex8 xa[MT];
        read_x(0, base, xa);
        ex8 af0 = convert(0, 0, cur, c);
        half(Flag<0>{}, base, cur, c, af0, xa);
        if constexpr (BAR) {
          W4A16_DMA_WAIT(NXT ? ND + WL : WL);
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (DMA) dma_x(b + 3, (b + 3) & 3);
          __builtin_amdgcn_sched_barrier(0);
        }
        read_x(4, base, xa);
        half(Flag<1>{}, base, cur, c, af0, xa);
      };
      // barrier 0, behind it the DMA for block 2
      W4A16_DMA_WAIT(ND + WL);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (n == 2) {  // no block 2: both slabs were requested in the prologue
        block_b(Flag<1>{}, Flag<0>{}, Flag<0>{}, 0, w0, w1);
        block_b(Flag<0>{}, Flag<0>{}, Flag<0>{}, 1, w1, w0);
      } else {
        dma_x(2, 2);
        __builtin_amdgcn_sched_barrier(0);
        // blocks 0 .. n - 4 have a block three ahead to fetch; then b = n - 3 (no DMA), n - 2 (no DMA), n - 1 (no barrier)
        int b = 0;
        for (; b + 2 <= n - 3; b += 2) {
          block_b(Flag<1>{}, Flag<1>{}, Flag<1>{}, b, w0, w1);
          block_b(Flag<1>{}, Flag<1>{}, Flag<1>{}, b + 1, w1, w0);
        }
        if (b < n - 3) {
          block_b(Flag<1>{}, Flag<1>{}, Flag<1>{}, b, w0, w1);
          w0 = w1;
          ++b;
        }
        block_b(Flag<1>{}, Flag<1>{}, Flag<0>{}, b, w0, w1);      // b = n - 3
        block_b(Flag<1>{}, Flag<0>{}, Flag<0>{}, b + 1, w1, w0);  // b = n - 2
        block_b(Flag<0>{}, Flag<0>{}, Flag<0>{}, b + 2, w0, w1);  // b = n - 1
      }
    }
  }

  STAMP(26);
#ifndef W4A16_SHARED_TSTORE
#define W4A16_SHARED_TSTORE 1
#endif
  if constexpr (!ACT && W4A16_SHARED_TSTORE) {
    // Stores through a transposing stage in LDS.  In the accumulators a lane holds 4 neighbouring columns of one row
    // per register group, so a store instruction straight from them writes 32 separate 16-byte pieces (two lanes per
    // row), and the launch's stores, which all workgroups of a round issue at the same time, are bound by the number
    // of write requests the L2 channels take, not by bytes.  Each wave instead writes a tile (32 rows x 64 bytes,
    // 16-byte chunks XOR-swizzled by the row) into 2 KiB of its own in the ring slab that is free by now (slab
    // KB % RING: no block KB was ever requested, and the last blocks read the other slabs), and reads it back as
    // 16 rows x 64 bytes per instruction: four lanes per row, 16 bytes each, a quarter of the write requests.
    // A wave's LDS operations execute in order, so no barrier is needed: the stage is private to the wave.
    unsigned char* const stage = smem_raw + (n % RING) * SLAB + w * 2048;
    const int wr = r32 * 64 + 8 * h, wsw = (r32 >> 1) & 3;
    const int rrow = lane >> 2, rch = lane & 3;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<ex4*>(stage + wr + ((q ^ wsw) << 4)) =
              AT::round(f32x4{acc[t][mt][4 * q], acc[t][mt][4 * q + 1], acc[t][mt][4 * q + 2], acc[t][mt][4 * q + 3]});
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          const int row = 16 * pass + rrow;
          const ex8 hv = *reinterpret_cast<const ex8*>(stage + row * 64 + ((rch ^ ((row >> 1) & 3)) << 4));
          const int m = m0 + mt * 32 + row, nn = (ct0 + t) * 32 + 8 * rch;
          if (m < a.M && nn < a.N) {  // (N is a multiple of 8: a chunk is inside or outside as a whole)
            ex4 lo = __builtin_shufflevector(hv, hv, 0, 1, 2, 3), hi = __builtin_shufflevector(hv, hv, 4, 5, 6, 7);
            if (a.bias != nullptr) {
              const elem* bias = reinterpret_cast<const elem*>(a.bias) + nn;
              lo = AT::add(lo, *reinterpret_cast<const ex4*>(bias));
              hi = AT::add(hi, *reinterpret_cast<const ex4*>(bias + 4));
            }
            ex8* dst = reinterpret_cast<ex8*>(reinterpret_cast<elem*>(a.out) + (size_t)m * a.N + nn);
            if (a.add_out) {  // fp16 + fp16 (bf16 + bf16, rounded once); one lane owns each element
              const ex8 old = *dst;
              lo = AT::add(__builtin_shufflevector(old, old, 0, 1, 2, 3), lo);
              hi = AT::add(__builtin_shufflevector(old, old, 4, 5, 6, 7), hi);
            }
            *dst = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
        }
      }
    STAMP(27);
#ifdef W4A16_STAMPS
    if (lane == 0 && STAMP_OK) g_stamps[((size_t)blockIdx.x * NWAVES + w) * 32 + 31] = __builtin_amdgcn_s_memrealtime();
#endif
  } else {
  static_assert(std::is_same<AT, ActF16>::value, "fp16 only from here");
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
  STAMP(27);
#ifdef W4A16_STAMPS
  if (lane == 0 && STAMP_OK) g_stamps[((size_t)blockIdx.x * NWAVES + w) * 32 + 31] = __builtin_amdgcn_s_memrealtime();
#endif
  }
#endif
}

template <int MT, int T, int NW>
int launch_wide_shared(const GemmArgs& a, hipStream_t s, int batch, bool act, bool bf16) {
  constexpr int lds = ring_slabs<MT, T, NW>() * MT * 32 * 256;  // ring of slabs
  static_assert(lds <= 160 * 1024, "LDS budget");
  // whole 128-k blocks, at least two of them, and 32-bit byte offsets
  if (a.K % 128 || a.K < 256 || (long long)a.NT * a.KBS * 1024 >= (1LL << 32) || (long long)a.M * a.K * 2 + kDmaImmBias >= (1LL << 32))
    return W4A16_E_SHAPE;
  if (act && a.N % 32) return W4A16_E_SHAPE;
  const int n_slices = ceil_div(ceil_div(a.N, 32), T * NW);
  const int m_blocks = ceil_div(a.M, MT * 32);
  const int blocks = ceil_div((long long)n_slices * m_blocks, 8) * 8;
  void (*kern)(int, int, int, const GemmArgs) = nullptr;
  if (bf16) {
    // bf16 activations: built for a few shapes, plain GEMM (+ bias, + add-to-out)
    if constexpr (wide_shared_bf16_built(MT, T, NW)) {
      if (act || batch != 1) return W4A16_E_SHAPE;
      if (a.G % 128 == 0)
        kern = gemm_wide_shared_kernel<MT, T, NW, 4, false, ActBf16>;
      else if (a.G == 64)
        kern = gemm_wide_shared_kernel<MT, T, NW, 2, false, ActBf16>;
      else
        kern = gemm_wide_shared_kernel<MT, T, NW, 1, false, ActBf16>;
    } else {
      return W4A16_E_SHAPE;
    }
  } else if (act) {
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
      w4a16_note_hip_error((int)e);
      return W4A16_E_LAUNCH;
    }
  }
  const int panel_rows = m_blocks > 8 ? kPanelRows : m_blocks;
  GemmArgs am = a;
  am.mb_magic = row_block_magic((long long)n_slices * m_blocks, m_blocks, panel_rows);
  hipLaunchKernelGGL(kern, dim3(blocks, 1, batch), dim3(64 * NW), lds, s, n_slices, m_blocks, panel_rows, am);
  return launched();
}


}  // namespace

int dispatch_wide_shared(const GemmArgs& a, int mt, int t, int nw, hipStream_t s, int batch, bool act, bool bf16) {
#define X(MT_, T_, NW_) \
  if (mt == MT_ && t == T_ && nw == NW_) return launch_wide_shared<MT_, T_, NW_>(a, s, batch, act, bf16);
  W4A16_FOR_EACH_WIDE_SHARED(X)
#undef X
  return W4A16_E_SHAPE;
}


}  // namespace w4a16

#ifdef W4A16_CHECK_DMA
// Checking build: reads {fragments that overtook their DMA, skip-waits switch}; `set_skip` >= 0 writes the switch and
// clears the count (synchronises).
extern "C" int w4a16_dbg_dma_state_shared(unsigned* host_state, int set_skip) {
  if (set_skip >= 0) {
    const unsigned v[2] = {0u, (unsigned)set_skip};
    const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(w4a16::g_dma_state), v, sizeof(v), 0, hipMemcpyHostToDevice);
    if (e != hipSuccess) return (int)e;
  }
  return (int)hipMemcpyFromSymbol(host_state, HIP_SYMBOL(w4a16::g_dma_state), 2 * sizeof(unsigned), 0, hipMemcpyDeviceToHost);
}
#endif

W4A16_STAMPS_ACCESSOR
