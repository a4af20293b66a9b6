// libw4a16.so: the 32x32x16 GEMM kernel with private activation slabs (K split over the waves of a workgroup)
// and its launcher.  Layout, arithmetic and the kernel families are described at the top of w4a16.hip.

#include "w4a16_common.h"

#ifdef W4A16_CHECK_DMA
namespace W4A16_HIDDEN w4a16 {
W4A16_DMA_STATE_DECL
}
#endif

namespace W4A16_HIDDEN w4a16 {
namespace {

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
// AT: the activation / output type (ActF16; ActBf16 for the bf16 entry point)
template <int MT, int T, int KW, int SZS, bool ACT = false, class AT = ActF16>
__global__ void __launch_bounds__(64 * KW) gemm_wide_kernel(int n_slices, int m_blocks, int panel_rows,
                                                            const GemmArgs ga, const WideSplit sp) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub (and has no buffer-resource type)
  typedef typename AT::elem elem;
  typedef typename AT::x4 ex4;
  typedef typename AT::x8 ex8;
  typedef typename AT::consts SZc;
  fetch_args_now(ga);
  asm volatile("" ::"s"(n_slices), "s"(m_blocks), "s"(panel_rows), "s"(sp.ws), "s"(sp.tickets), "s"(sp.kb_per_split));
  if (ga.mask != nullptr && ga.mask[blockIdx.z] == 0) return;
  const GemmArgs a = batch_slice(ga);
  constexpr int R = MT * 32;             // rows of the workgroup's tile
  constexpr int HALF = R * 128;          // bytes of a half slab: R rows x 64 k
  constexpr int SLAB = 2 * HALF;         // LDS per wave
  constexpr int XLR = R / 8;             // DMA loads per half slab (8 rows x 128 bytes each)
  constexpr int XL = (W4A16_DBG & 1) ? 1 : XLR;
  constexpr int NSZ = 4 / SZS;           // scale/zero words per 128-k block and tile
  constexpr int WL = T * (NSZ + 2);      // register loads per weight block
  // 4-wave shapes with two row tiles keep TWO whole slabs per wave (4 x 32 KiB of LDS) and fetch the next block's
  // slab piece by piece, MT pieces per k-step, between the matrix instructions of the current block (see block2)
  constexpr bool DBL = wide_double_slab(MT, KW) && !(W4A16_DBG & 1);
  constexpr int WSLABS = DBL ? 2 : 1;    // slabs per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r32 = lane & 31, h = lane >> 5;
#ifdef W4A16_CHECK_DMA
  const bool dma_skip_waits = __builtin_amdgcn_readfirstlane(g_dma_state[1]) != 0;
#endif
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
  STAMP(24);
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
    xoff[p] = ((uint32_t)min(m0 + 8 * p + (lane >> 3), a.M - 1) * a.K + xchunk[p & 1] * 8) * 2u + kDmaImmBias;
  const int wbase = w * (WSLABS * SLAB);
  // Set up AFTER the wave's first requests are out (see below): the LDS addresses of the B fragments, the
  // accumulators and the conversion's constants.
  int xrd[4];
  f32x16 acc[T][MT];
  DqConst dqc;
  STAMP(25);

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
  // (the activations' descriptor starts kDmaImmBias bytes BEFORE the matrix and every lane offset is larger by that
  // much: a DMA instruction with an immediate offset then takes it off its lane offset without going below zero)
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<unsigned char*>(const_cast<f16*>(a.x)) - kDmaImmBias, 0, (int)((uint32_t)a.M * (uint32_t)a.K * 2u + kDmaImmBias), 0x00020000);
  auto load_w = [&](int kb, WSet& b) {
    if (W4A16_DBG & 2) kb = w;  // ablation: weights from cache
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
  // DMA piece p (0 .. XLR - 1) of half hp of block kb's slab -> LDS half hp of this wave's slab `sl` (K is a multiple
  // of 128 here)
  auto dma_piece = [&](int kb, int hp, int p, int sl) {
    const int dst = wbase + sl * SLAB + hp * HALF + p * 1024;
#ifdef W4A16_CHECK_DMA
    *reinterpret_cast<u32x4*>(smem_raw + dst + lane * 16) = u32x4{kDmaPoison, kDmaPoison, kDmaPoison, kDmaPoison};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the pattern is in place before the DMA can land on it
#endif
    // The instruction's immediate offset (up to 4095) moves the LDS address AND the source address: pieces 4 j .. 4 j + 3
    // share one LDS base (M0 is then written once per four pieces), their lane offsets are smaller by the same amount.
    auto issue = [&](auto imm_) {  // (the immediate has to be a literal: p is a constant only after unrolling)
      constexpr int IMM = decltype(imm_)::value;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          x_rs, (__attribute__((address_space(3))) void*)(smem_raw + dst - IMM), 16, xoff[p] - (uint32_t)IMM,
          (kb * 128 + hp * 64) * 2, IMM, 0);
    };
    switch (p & 3) {
      case 0: issue(Flag<0>{}); break;
      case 1: issue(Flag<1024>{}); break;
      case 2: issue(Flag<2048>{}); break;
      default: issue(Flag<3072>{}); break;
    }
#ifdef W4A16_CHECK_DMA
    if (dma_skip_waits && p == 0) dma_check(*reinterpret_cast<const u32x4*>(smem_raw + dst + lane * 16), &g_dma_state[0]);
#endif
  };
  // XL DMA loads: a whole half
  auto dma_x = [&](int kb, int hp, int sl = 0) {
#pragma unroll
    for (int p = 0; p < XL; ++p) dma_piece(kb, hp, p, sl);  // (ablation 1: XL is 1, an eighth of the activation traffic)
  };
  // B fragments of k-step `kstep` (0..7) of the block: 2 ds_read_b128 from half kstep / 4
  auto read_x = [&](int kstep, ex8 (&xf)[MT], int sl = 0) {
    const int hp = kstep >> 2, ks = kstep & 3;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      xf[mt] = *reinterpret_cast<const ex8*>(smem_raw + xrd[ks] + sl * SLAB + hp * HALF + mt * 4096);
#ifdef W4A16_CHECK_DMA
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      dma_check(__builtin_bit_cast(u32x4, xf[mt]), &g_dma_state[0]);
#endif
  };
  // The weight fragment of tile t at k-step `kstep` (scale/zero constants refreshed where a new group starts)
  auto convert = [&](int kstep, int t, const WSet& b, SZc (&c)[T]) {
    if (kstep % (2 * SZS) == 0) c[t] = AT::expand(b.sz[t][kstep / (2 * SZS)]);
    if (W4A16_DBG & 4) return __builtin_bit_cast(ex8, b.q[t][kstep & 1]);  // ablation: no conversion
    return AT::dequant(b.q[t][kstep & 1][kstep >> 1], c[t], dqc);
  };
  // One k-step: MT x T matrix instructions.  The conversion runs ONE TILE AHEAD of them (`af0` comes in
  // converted; tiles 1.. and tile 0 of the next k-step are converted here), and the two instruction kinds
  // are dealt out alternately: a wave issues in order, so a matrix instruction that finds the pipe still
  // busy with its predecessor would otherwise hold back the ~52 cycles of conversion behind it, and a
  // single wave would run at half the matrix rate.
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
    {
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
    SZc c[T];
    ex8 xa[MT], xb[MT];
    if constexpr (!LAST) {
      load_w(kb_next, nxt);
      __builtin_amdgcn_sched_barrier(0);
    }
    W4A16_DMA_WAIT(LAST ? XL : XL + WL);
    read_x(0, xa);
    ex8 af0 = convert(0, 0, cur, c);
#pragma unroll
    for (int k2 = 0; k2 < 8; k2 += 2) {
      // even k-step on xa while xb arrives, odd k-step on xb while xa arrives
      read_x(k2 + 1, xb);
      __builtin_amdgcn_sched_barrier(0);
      math(k2, cur, c, xa, af0);
      __builtin_amdgcn_sched_barrier(0);
      if (k2 + 1 == 3) W4A16_DMA_WAIT(LAST ? 0 : WL);
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

  // One k-step for block2: the same arithmetic tile by tile, each tile's matrix instructions and the next tile's
  // conversion in a scheduling region of their own, and `between(t)` (a DMA piece) issued in program order behind
  // tile t's region: right after a matrix instruction has started, so the piece goes out in its shadow.
  auto math2 = [&](int kstep, const WSet& b, SZc (&c)[T], const ex8 (&xf)[MT], ex8& af0, auto&& between) {
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
      for (int i = 0; i < MT; ++i) {  // (the matrix instruction last: `between` then goes out in its shadow)
        __builtin_amdgcn_sched_group_barrier(0x002, per, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      af = nx;
      __builtin_amdgcn_sched_barrier(0);
      between(t);
      __builtin_amdgcn_sched_barrier(0);
    }
    af0 = af;
  };

  // The same with two whole slabs per wave (DBL): block i reads slab i & 1 while the slab of the wave's next block
  // arrives in the other one, MT pieces per k-step, each issued between the matrix instructions of that k-step: a
  // burst of 8 DMA instructions from each of the four waves at once holds a wave's in-order issue for ~60 cycles per
  // instruction (the CU's address path takes one 1-KiB instruction per ~16 cycles), a third of a block's time; one
  // at a time they go out in the shadow of a matrix instruction.  Issue order per block: W(next), then the 2 XLR
  // pieces (half 0 in k-steps 0-3, half 1 in 4-7).  Half 0 of the current slab was requested in k-steps 0-3 of the
  // previous block: at the start all but the XLR + WL youngest operations must be done (XLR in the last block);
  // half 1 is first read after k-step 2, when 3 MT pieces of this block are out: all but the 3 MT + WL youngest
  // (none in the last block).
  auto block2 = [&](auto last_, auto sl_, int kb_next, const WSet& cur, WSet& nxt) {
    constexpr bool LAST = decltype(last_)::value != 0;
    constexpr int SL = decltype(sl_)::value;
    SZc c[T];
    ex8 xa[MT], xb[MT];
    if constexpr (!LAST) {
      load_w(kb_next, nxt);
      __builtin_amdgcn_sched_barrier(0);
    }
    W4A16_DMA_WAIT(LAST ? XLR : XLR + WL);
    read_x(0, xa, SL);
    ex8 af0 = convert(0, 0, cur, c);
    // piece j (0 .. MT - 1) of k-step `kstep`: pieces 0 .. XLR - 1 fill half 0, the rest half 1
    auto piece = [&](int kstep, int j) {
      if constexpr (!LAST) {
        if (j < MT) {
          const int q = kstep * MT + j;
          dma_piece(kb_next, q / XLR, q % XLR, SL ^ 1);
        }
      }
    };
#pragma unroll
    for (int k2 = 0; k2 < 8; k2 += 2) {
      read_x(k2 + 1, xb, SL);
      __builtin_amdgcn_sched_barrier(0);
      math2(k2, cur, c, xa, af0, [&](int t) { piece(k2, t); });
      if (k2 + 1 == 3) W4A16_DMA_WAIT(LAST ? 0 : WL + 3 * MT);
      if (k2 + 2 < 8) read_x(k2 + 2, xa, SL);
      __builtin_amdgcn_sched_barrier(0);
      math2(k2 + 1, cur, c, xb, af0, [&](int t) { piece(k2 + 1, t); });
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
  // With gridDim.y > 1 the workgroups of an output tile split K between them: this one takes the 128-k blocks
  // [kb_lo, kb_hi) (the host leaves no split empty).
  constexpr bool HEAD = (KW == 8);
  const int kb_lo = blockIdx.y * sp.kb_per_split;
  const int kb_hi = min(kb_lo + sp.kb_per_split, a.KB);
  auto next = [&](int kb) { return (HEAD && kb - kb_lo < 8) ? kb + 4 : kb + KW; };
  int kb = kb_lo + ((HEAD && w >= 4) ? 8 + w : w);
  int nblk = 0;
  if constexpr (HEAD) {
    for (int b = kb; b < kb_hi; b = next(b)) ++nblk;
  } else {
    nblk = kb < kb_hi ? (kb_hi - kb + KW - 1) / KW : 0;
  }
  WSet w0, w1;
  if (nblk > 0) {
    load_w(kb, w0);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(26);
    const int sl0 = DBL ? (nblk & 1) : 0;  // (two slabs per wave: see the main loop)
    dma_x(kb, 0, sl0);
    __builtin_amdgcn_sched_barrier(0);  // half 0 before half 1: the counted waits rely on the issue order
    STAMP(27);
    dma_x(kb, 1, sl0);
    __builtin_amdgcn_sched_barrier(0);
  }
  // (everything the first block needs is on its way: now the set-up that needs no memory)
  {
    const int f = (r32 >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) xrd[ks] = wbase + r32 * 128 + (((2 * ks + h) ^ f) << 4);  // row r32 (+ 32 mt), chunk 2 ks + h
  }
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][mt][i] = 0.f;
  dqc = make_dq_const();
  __builtin_amdgcn_sched_barrier(0);
  STAMP(1);
  if (nblk > 0) {
    int i = 0;
    if constexpr (DBL) {
      // Blocks alternate between the two slabs at compile time, two per trip; the wave's last block always reads
      // slab 1 and weight set w0 (one call site: the accumulators never meet from two paths).  So the prologue
      // fills slab 1 when the number of blocks is odd, and a leading single block brings the rest to an even count.
      if ((nblk & 1) && nblk > 1) {
        kb = next(kb);
        block2(Flag<0>{}, Flag<1>{}, kb, w0, w1);
        w0 = w1;
        i = 1;
      }
      for (; i + 2 < nblk; i += 2) {
        const int k1 = next(kb), k2 = next(k1);
        block2(Flag<0>{}, Flag<0>{}, k1, w0, w1);
        block2(Flag<0>{}, Flag<1>{}, k2, w1, w0);
        kb = k2;
      }
      if (nblk > 1) {
        kb = next(kb);
        block2(Flag<0>{}, Flag<0>{}, kb, w0, w1);
        w0 = w1;
      }
      block2(Flag<1>{}, Flag<1>{}, kb, w0, w1);
    } else {
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
  }

  // Cross-wave reduction through LDS (the slabs are free once every wave is past the first barrier).  A wave's
  // partial result is T * MT * 4 slices of 1 KiB: 4 accumulator registers x 64 lanes, i.e. for tile (t, mt) and
  // register group q the outputs of token 32 mt + r32, columns 32 t + 8 q + 4 h .. + 3.
  //  * 8 waves: the upper four park their slices, the lower four add them to their accumulators and carry on alone.
  //  * 4 waves: every slice has an OWNER wave that keeps its own share in registers; each wave leaves the slices it
  //    does not own in their owners' inboxes (three quarters of the partial sums travel through LDS, once), then
  //    adds the three inbox copies to each slice it owns, in wave order.  The owner of slice (tile, q) is wave q;
  //    with the fused activation wave 2 (tile & 1) + (q & 1), so that the gate (q = 0, 1) and up (q = 2, 3) halves
  //    of an output meet in one wave.
  constexpr int TM = T * MT;
  constexpr int NSL = TM * 4;
  constexpr int PER = wide_slices_per_owner(MT, T, ACT);  // (with the fused activation and an odd tile count two owners hold a pair fewer)
  static_assert(KW == 4 || KW == 8, "four partial sums meet in the final exchange");
  static_assert(12 * PER * 1024 <= 160 * 1024, "inboxes: 4 owners x 3 senders x PER KiB");
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw);
  auto part = [&](int i) {
    const int t = i / (MT * 4), mt = (i >> 2) % MT, q = i & 3;
    return f32x4{acc[t][mt][4 * q], acc[t][mt][4 * q + 1], acc[t][mt][4 * q + 2], acc[t][mt][4 * q + 3]};
  };
  STAMP(20);
  __syncthreads();
  STAMP(21);
  // 8 waves whose slices divide evenly among them (and whose 8 x 7 inboxes fit): one exchange among all eight, every
  // wave owns NSL / 8 slices (slice i belongs to wave i & 7), sums the eight partial results in wave order and stores
  // them.  One barrier and 7 slices each way per owned slice, where parking and the four-owner exchange take three
  // barriers and twice the LDS traffic on the critical path.
  #ifndef W4A16_DIRECT8
#define W4A16_DIRECT8 1
#endif
  // (Four slices, one 32 x 32 tile: waves 0-3 own one each, all eight send.)
  constexpr bool DIRECT8 = W4A16_DIRECT8 && KW == 8 && !ACT && (NSL % 8 == 0 || NSL == 4) && 56 * ((NSL + 7) / 8) <= 160;
  if constexpr (DIRECT8) {
    constexpr int P8 = (NSL + 7) / 8;
    f32x4 fin8[P8];
    auto exchange8 = [&](auto me_) {
      constexpr int ME = decltype(me_)::value;
#pragma unroll
      for (int i = 0; i < NSL; ++i) {
        const int o = i & 7;
        if (o != ME) red[((o * 7 + ((ME - o - 1) & 7)) * P8 + (i >> 3)) * 64 + lane] = part(i);
      }
      __syncthreads();
      STAMP(22);
      if constexpr (ME >= NSL) return;  // (owns nothing)
      // inbox f of owner ME holds the copy of wave (ME + 1 + f) & 7: all reads first, then the sums in wave order
      f32x4 got[P8][7];
#pragma unroll
      for (int j = 0; j < P8; ++j)
#pragma unroll
        for (int f = 0; f < 7; ++f) got[j][f] = red[((ME * 7 + f) * P8 + j) * 64 + lane];
#pragma unroll
      for (int j = 0; j < P8; ++j) {
        fin8[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int src = 0; src < 8; ++src) fin8[j] += src == ME ? part(8 * j + ME) : got[j][(src - ME - 1) & 7];
      }
    };
    switch (w) {
      case 0: exchange8(Flag<0>{}); break;
      case 1: exchange8(Flag<1>{}); break;
      case 2: exchange8(Flag<2>{}); break;
      case 3: exchange8(Flag<3>{}); break;
      case 4: exchange8(Flag<4>{}); break;
      case 5: exchange8(Flag<5>{}); break;
      case 6: exchange8(Flag<6>{}); break;
      default: exchange8(Flag<7>{}); break;
    }
    if (w >= NSL) return;  // (four slices: the upper waves own nothing)
#pragma unroll
    for (int j = 0; j < P8; ++j) {
      const int tm = 2 * j + (w >> 2), q = w & 3;  // slice 8 j + w = (tile tm, register group q)
      const int t = tm / MT, mt = tm % MT;
      const int m = m0 + mt * 32 + r32;
      const int n = (ct0 + t) * 32 + 8 * q + 4 * h;
      if (m < a.M && n < a.N) {
        ex4 hv = AT::round(fin8[j]);
        if (a.bias != nullptr) hv = AT::add(hv, *reinterpret_cast<const ex4*>(reinterpret_cast<const elem*>(a.bias) + n));
        ex4* dst = reinterpret_cast<ex4*>(reinterpret_cast<elem*>(a.out) + (size_t)m * a.N + n);
        if (a.add_out) hv = AT::add(*dst, hv);
        *dst = hv;
      }
    }
    STAMP(23);
#ifdef W4A16_STAMPS
    if (lane == 0 && STAMP_OK) g_stamps[((size_t)blockIdx.x * NWAVES + w) * 32 + 31] = __builtin_amdgcn_s_memrealtime();
#endif
    return;
  }
  if constexpr (KW == 8) {
    if (w >= 4) {
#pragma unroll
      for (int i = 0; i < NSL; ++i) red[((w - 4) * NSL + i) * 64 + lane] = part(i);
    }
    __syncthreads();
    if (w >= 4) return;  // (the barriers below count the waves that are still running)
    {
      // all reads first, then the sums: one round trip to LDS, not NSL of them
      f32x4 other[NSL];
#pragma unroll
      for (int i = 0; i < NSL; ++i) other[i] = red[(w * NSL + i) * 64 + lane];
#pragma unroll
      for (int i = 0; i < NSL; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i / (MT * 4)][(i >> 2) % MT][4 * (i & 3) + e] += other[i][e];
    }
    __syncthreads();  // the parked slices have been read: their LDS is free for the inboxes
  }
  // compile-time in the unrolled loops below: owner of slice i and its position among the owner's slices
  auto owner_of = [](int i) { return ACT ? 2 * ((i >> 2) & 1) + (i & 1) : (i & 3); };
  auto pos_of = [](int i) { return ACT ? 2 * (i >> 3) + ((i >> 1) & 1) : (i >> 2); };
  // fin[j]: the finished fp32 sums of the j-th slice this wave owns.  The accumulators can only be indexed with
  // compile-time constants, so the body is instantiated once per wave number.
  f32x4 fin[PER];
  auto exchange = [&](auto me_) {
    constexpr int ME = decltype(me_)::value;
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
      const int o = owner_of(i);
      if (o != ME) red[((o * 3 + ((ME - o - 1) & 3)) * PER + pos_of(i)) * 64 + lane] = part(i);
    }
    __syncthreads();
    STAMP(22);
    // inbox f of owner ME holds the copy of wave (ME + 1 + f) & 3: summed in wave order 0, 1, 2, 3
#pragma unroll
    for (int j = 0; j < PER; ++j) fin[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int src = 0; src < 4; ++src) {
#pragma unroll
      for (int i = 0; i < NSL; ++i) {
        if (owner_of(i) != ME) continue;
        const int j = pos_of(i);
        if (src == ME)
          fin[j] += part(i);
        else
          fin[j] += red[((ME * 3 + ((src - ME - 1) & 3)) * PER + j) * 64 + lane];
      }
    }
  };
  if (w == 0) exchange(Flag<0>{});
  else if (w == 1) exchange(Flag<1>{});
  else if (w == 2) exchange(Flag<2>{});
  else exchange(Flag<3>{});

  // K split over workgroups: every workgroup of an output tile leaves its finished partial sums in its own slot
  // of the workspace (whole 1-KiB wave stores that go through to memory, each 128-byte line written by one
  // instruction), waits for them to be acknowledged, and one lane draws a ticket.  The workgroup that draws the
  // last ticket of the tile reads the other slots back (loads that bypass this CU's cache), adds all splits in
  // split order, so the result does not depend on who came last, finishes the tile and returns the counter to
  // zero for the next launch.
  const int splits = gridDim.y;
  if (splits > 1) {
    const int tile = slice * m_blocks + mb;
    const uint32_t slot = blockIdx.z * (n_slices * m_blocks) + tile;
    constexpr uint32_t WGB = 4 * PER * 1024;  // bytes of one workgroup's partial sums
    const __amdgpu_buffer_rsrc_t ws_rs = __builtin_amdgcn_make_buffer_rsrc(sp.ws, 0, sp.ws_bytes, 0x00020000);
    const uint32_t lane_off = (uint32_t)(w * PER) * 1024u + lane * 16u;
    constexpr int THROUGH = 16;  // cache policy sc1: the store goes through to memory, the load bypasses this CU's cache
    const uint32_t mine = (slot * splits + blockIdx.y) * WGB;
#pragma unroll
    for (int j = 0; j < PER; ++j)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, fin[j]), ws_rs, lane_off + j * 1024, mine, THROUGH);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // every wave's stores are acknowledged before the ticket is drawn
    unsigned* const flag = reinterpret_cast<unsigned*>(smem_raw + 12 * PER * 1024);  // (just above the inboxes)
    if (threadIdx.x == 0)
      *flag = __hip_atomic_fetch_add(&sp.tickets[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();  // the wave that drew the ticket has its answer before anybody loads
    const unsigned old = __builtin_amdgcn_readfirstlane(*flag);
    if (old != (unsigned)(splits - 1)) return;  // whole workgroup
    for (int s2 = 0; s2 < splits; ++s2) {
      const uint32_t theirs = (slot * splits + s2) * WGB;
      f32x4 got[PER];
#pragma unroll
      for (int j = 0; j < PER; ++j)
        got[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ws_rs, lane_off + j * 1024, theirs, THROUGH));
#pragma unroll
      for (int j = 0; j < PER; ++j) fin[j] = s2 == 0 ? got[j] : fin[j] + got[j];
    }
    if (threadIdx.x == 0) __hip_atomic_store(&sp.tickets[slot], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // Epilogue of the slices this wave owns: rounded to fp16 first, then the bias is added in fp16, then the
  // activation: what the separate steps on an fp16 GEMM result give.
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    if constexpr (ACT) {
      // slices (tile, q) and (tile, q + 2): gate columns and the same columns of the up half
      if (j & 1) continue;
      const int tm = 2 * (j >> 1) + (w >> 1), q = w & 1;
      if (tm >= TM) continue;
      const int t = tm / MT, mt = tm % MT;
      const int m = m0 + mt * 32 + r32;
      const int NH = a.N / 2;
      const int n = (ct0 + t) * 16 + 8 * q + 4 * h;
      if (m < a.M && n < NH) {
        const f32x4 v = fin[j], vu = fin[j + 1];
        ex4 hg = AT::round(v);
        ex4 hu = AT::round(vu);
        if (a.bias != nullptr) {
          hg = AT::add(hg, *reinterpret_cast<const ex4*>(reinterpret_cast<const elem*>(a.bias) + n));
          hu = AT::add(hu, *reinterpret_cast<const ex4*>(reinterpret_cast<const elem*>(a.bias) + NH + n));
        }
        *reinterpret_cast<ex4*>(reinterpret_cast<elem*>(a.out) + (size_t)m * NH + n) = silu_mul(hg, hu);
      }
    } else {
      const int t = j / MT, mt = j % MT, q = w;
      const int m = m0 + mt * 32 + r32;
      const int n = (ct0 + t) * 32 + 8 * q + 4 * h;
      if (m < a.M && n < a.N) {
        const f32x4 v = fin[j];
        ex4 hv = AT::round(v);
        if (a.bias != nullptr) hv = AT::add(hv, *reinterpret_cast<const ex4*>(reinterpret_cast<const elem*>(a.bias) + n));
        ex4* dst = reinterpret_cast<ex4*>(reinterpret_cast<elem*>(a.out) + (size_t)m * a.N + n);
        if (a.add_out) hv = AT::add(*dst, hv);  // fp16 + fp16 (bf16 + bf16); one lane owns each element
        *dst = hv;
      }
    }
  }
  STAMP(23);
#ifdef W4A16_STAMPS
  if (lane == 0 && STAMP_OK) g_stamps[((size_t)blockIdx.x * NWAVES + w) * 32 + 31] = __builtin_amdgcn_s_memrealtime();
#endif
#endif
}

template <int MT, int T, int KW>
int launch_wide(const GemmArgs& a, hipStream_t s, int batch, bool act, int splits, void* ws, size_t ws_bytes, bool bf16 = false) {
  constexpr int stage = KW * MT * 32 * 256 * (wide_double_slab(MT, KW) ? 2 : 1);  // one or two private slabs per wave
  constexpr int per = wide_slices_per_owner(MT, T, true);  // (the larger of the two epilogues' inbox sizes)
  constexpr int inbox = 12 * per * 1024 + 16;               // 4 owners x 3 senders, then the ticket word
  constexpr int park = KW > 4 ? 4 * T * MT * 4 * 1024 : 0;  // 8 waves: the upper four park whole partial tiles first
  constexpr int direct8 = KW == 8 && ((T * MT * 4) % 8 == 0 || T * MT == 1) ? 56 * ((T * MT * 4 + 7) / 8) * 1024 : 0;  // or: 8 owners x 7 senders
  constexpr int reduce0 = park > inbox ? park : inbox;
  constexpr int reduce = direct8 > reduce0 && direct8 <= 160 * 1024 ? direct8 : reduce0;
  constexpr int lds = stage > reduce ? stage : reduce;
  static_assert(lds <= 160 * 1024, "LDS budget");
  // whole 128-k blocks only, and 32-bit byte offsets into the three buffers
  if (a.K % 128 || (long long)a.NT * a.KBS * 1024 >= (1LL << 32) || (long long)a.M * a.K * 2 + kDmaImmBias >= (1LL << 32)) return W4A16_E_SHAPE;
  // fused activation: a tile is 16 gate + 16 up columns, so there are N / 32 of them too
  if (act && a.N % 32) return W4A16_E_SHAPE;
  const int n_slices = ceil_div(ceil_div(a.N, 32), T);
  const int m_blocks = ceil_div(a.M, MT * 32);
  const int blocks = ceil_div((long long)n_slices * m_blocks, 8) * 8;
  // K split over workgroups: 4-wave shapes only; every split gets at least one block per wave
  if (splits < 1 || splits > kMaxSplits) return W4A16_E_SHAPE;
  WideSplit sp{nullptr, nullptr, 0, a.KB};
  if (splits > 1) {
    if (KW != 4) return W4A16_E_SHAPE;
    sp.kb_per_split = ceil_div(a.KB, splits);
    splits = ceil_div(a.KB, sp.kb_per_split);  // no empty split
  }
  if (splits > 1) {
    if (!ws) return W4A16_E_NULL;
    if (misaligned(ws)) return W4A16_E_ALIGN;
    const long long tiles = (long long)batch * n_slices * m_blocks;
    const size_t need = wide_ws_bytes(tiles, splits, MT, T, act);
    if (tiles > kMaxTickets || need - kTicketBytes >= (1ULL << 31)) return W4A16_E_SHAPE;
    if (ws_bytes < need) return W4A16_E_WORKSPACE;
    sp.tickets = static_cast<unsigned*>(ws);
    sp.ws = reinterpret_cast<float*>(static_cast<unsigned char*>(ws) + kTicketBytes);
    sp.ws_bytes = (int)(need - kTicketBytes);
  }
  void (*kern)(int, int, int, const GemmArgs, const WideSplit) = nullptr;
  if (bf16) {
    // bf16 activations: built for a few shapes, plain GEMM (+ bias, + add-to-out)
    if constexpr (wide_bf16_built(MT, T, KW)) {
      if (act || splits > 1 || batch != 1) return W4A16_E_SHAPE;
      if (a.G % 128 == 0)
        kern = gemm_wide_kernel<MT, T, KW, 4, false, ActBf16>;
      else if (a.G == 64)
        kern = gemm_wide_kernel<MT, T, KW, 2, false, ActBf16>;
      else
        kern = gemm_wide_kernel<MT, T, KW, 1, false, ActBf16>;
    } else {
      return W4A16_E_SHAPE;
    }
  } else if (act) {
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
      w4a16_note_hip_error((int)e);
      return W4A16_E_LAUNCH;
    }
  }
  const int panel_rows = m_blocks > 8 ? kPanelRows : m_blocks;
  GemmArgs am = a;
  am.mb_magic = row_block_magic((long long)n_slices * m_blocks, m_blocks, panel_rows);
  hipLaunchKernelGGL(kern, dim3(blocks, splits, batch), dim3(64 * KW), lds, s, n_slices, m_blocks, panel_rows, am, sp);
  return launched();
}

}  // namespace

int dispatch_wide(const GemmArgs& a, int mt, int t, int kw, hipStream_t s, int batch, bool act, int splits, void* ws,
                  size_t ws_bytes, bool bf16) {
#define X(MT_, T_, KW_) \
  if (mt == MT_ && t == T_ && kw == KW_) return launch_wide<MT_, T_, KW_>(a, s, batch, act, splits, ws, ws_bytes, bf16);
  W4A16_FOR_EACH_WIDE(X)
#undef X
  return W4A16_E_SHAPE;
}

}  // namespace w4a16

W4A16_STAMPS_ACCESSOR

#ifdef W4A16_CHECK_DMA
// Checking build: reads {fragments that overtook their DMA, skip-waits switch}; `set_skip` >= 0 writes the switch and
// clears the count (synchronises).
extern "C" int w4a16_dbg_dma_state_wide(unsigned* host_state, int set_skip) {
  if (set_skip >= 0) {
    const unsigned v[2] = {0u, (unsigned)set_skip};
    const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(w4a16::g_dma_state), v, sizeof(v), 0, hipMemcpyHostToDevice);
    if (e != hipSuccess) return (int)e;
  }
  return (int)hipMemcpyFromSymbol(host_state, HIP_SYMBOL(w4a16::g_dma_state), 2 * sizeof(unsigned), 0, hipMemcpyDeviceToHost);
}
#endif
