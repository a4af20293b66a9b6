// libw4a16.so: the decode GEMM kernel (up to 16 rows, paced weight stream, optional K split over workgroups)
// and its launcher.  Layout, arithmetic and the kernel families are described at the top of w4a16.hip.

#include "w4a16_common.h"

namespace W4A16_HIDDEN w4a16 {
namespace {

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
// XLDS = false serves 5-16 rows: the activations then come as B fragments straight from L2, one block ahead of
// the block being worked on (at 16 rows a fragment load has no lane to spare, and 16 rows x 256 bytes per block
// and wave would not fit LDS), the waves park whole 16 x 16 partial tiles, and waves 0..H-1 each sum one tile.
// AT: the activation / output type (fp16 or bf16; the weights of the bf16 path are bf16((q - z) * s), see
// dequant_word_bf16).
// XM: how the activations reach the matrix instruction.  1: up to 4 rows, all of a wave's blocks staged in LDS up
// front (above).  0: 5-16 rows, B fragments straight from L2.  2: 5-16 rows through a ring of DB slots of 4 KiB per
// wave: a block's 16 rows x 256 bytes arrive by four LDS-DMA instructions of 4 rows x 256 contiguous bytes (whole
// cache lines, where a fragment load touches 16 separate 64-byte pieces; rows past M - 1 repeat the last row), chunk
// c of row r at 16-byte position c ^ r of the row, so that the ds_read_b128 fragment reads are conflict-free; the
// slot of block b is refilled for block b + DB right after the block's first unit has consumed its fragments.
// 3: 17-32 rows, the ring with two 16-row tiles (slots of 8 KiB, eight DMA instructions per block): a converted weight
// fragment feeds two matrix instructions.
template <int T, int W, int DB, int SZS, bool ACT = false, int XM = 1, class AT = ActF16>
__global__ void __launch_bounds__(64 * W) gemm_decode_kernel(const GemmArgs ga, const DecodeArgs e) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub (and has no buffer-resource type)
  constexpr bool XLDS = XM == 1, XRING = XM >= 2;
  constexpr int MT = XM == 3 ? 2 : 1;  // 16-row tiles
  static_assert(MT == 1 || !ACT, "two row tiles: plain GEMM only");
  constexpr int SLOT = MT * 4096;      // bytes of a ring slot
  typedef typename AT::elem elem;
  typedef typename AT::x4 ex4;
  typedef typename AT::x8 ex8;
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
  if (e.steal) {
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
  const int xwave = XRING ? w * (DB * SLOT) : w * nb_max * 1024;
  // B fragment of k-step j: row min(n16, M - 1) (the matrix instruction's columns past M - 1 are never stored),
  // chunk 4 j + kq
  int xrd[4];
  {
    const int rr = min(n16, a.M - 1);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      xrd[j] = XRING ? xwave + n16 * 256 + (((4 * j + kq) ^ n16) << 4) : xwave + rr * 256 + ((4 * (j ^ rr) + kq) << 4);
  }
  // ring: DMA i of a block fills rows 4 i .. 4 i + 3: lane -> row 4 i + (lane >> 4), position lane & 15
  constexpr int XD = 4 * MT;  // DMA instructions per block
  uint32_t xring[XRING ? XD : 1];
  if constexpr (XRING) {
#pragma unroll
    for (int i = 0; i < XD; ++i) {
      const int row = 4 * i + kq;
      xring[i] = ((uint32_t)min(row, a.M - 1) * a.K + ((n16 ^ (row & 15)) * 8)) * 2u;
    }
  }
  auto dma_ring = [&](int kb, int slot) {
    if constexpr (XRING) {
#pragma unroll
      for (int i = 0; i < XD; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr)(smem_raw + xwave + slot * SLOT + i * 1024), 16, xring[i],
                                                 kb * 256, 0, 0);
    }
  };

  f32x4 acc[T][MT];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
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
  auto compute_unit = [&](int t, const Unit& d, const ex8 (&xf)[4][MT]) {
    typename AT::consts c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j % SZS == 0) c = AT::expand(d.sz[j / SZS]);
      const ex8 af = AT::dequant(d.q[j], c, dqc);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[t][mt] = AT::mfma16(af, xf[j][mt], acc[t][mt]);
    }
  };
  // Without LDS staging: the B fragments of a block (row min(n16, M - 1), k = 32 j + 8 kq .. + 7), 4 loads
  const uint32_t xfrag_lane = ((uint32_t)min(n16, a.M - 1) * a.K + kq * 8) * 2u;
  auto load_x = [&](int kb, ex8 (&xf)[4][MT]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      xf[j][0] = __builtin_bit_cast(ex8, __builtin_amdgcn_raw_buffer_load_b128(x_rs, xfrag_lane, (kb * 128 + j * 32) * 2, 0));
  };
  // One block from the registers `u`; with REFILL each unit's registers are asked to fetch the same tile's unit
  // of block kb_refill as soon as the unit is consumed; with XNEXT (fragments from L2 only) the next block's
  // fragments are requested into `xnext` first.  (Compile-time flags: a load under a run-time branch makes the
  // compiler wait for every load in flight.)
  auto step = [&](auto refill_, auto xnext_, int bi, Unit (&u)[T], ex8 (&xcur)[4][MT], ex8 (&xnext)[4][MT], int kb_refill,
                  int kb_xnext) {
    constexpr bool REFILL = decltype(refill_)::value != 0;
    constexpr bool XNEXT = decltype(xnext_)::value != 0;
    ex8 xl[4][MT];
    if constexpr (XLDS) {
#pragma unroll
      for (int j = 0; j < 4; ++j) xl[j][0] = *reinterpret_cast<const ex8*>(smem_raw + xrd[j] + bi * 1024);
    } else if constexpr (XRING) {
      // The slot's DMA was issued ahead of this block's weights; younger than it are those T * UL loads and, with two
      // blocks in flight and a block after this one, that block's DMA and weights.  (The counter has 6 bits.)
      constexpr int YOUNGER = (DB == 2 && XNEXT) ? 2 * T * UL + XD : T * UL;
      wait_vmcnt<(YOUNGER < 63 ? YOUNGER : 63)>();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)  // (row 16 mt + n16 lies 4096 bytes on, swizzled alike)
          xl[j][mt] = *reinterpret_cast<const ex8*>(smem_raw + xrd[j] + (bi % DB) * SLOT + mt * 4096);
    } else if constexpr (XNEXT) {
      load_x(kb_xnext, xnext);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if constexpr (XLDS || XRING)
        compute_unit(t, u[t], xl);
      else
        compute_unit(t, u[t], xcur);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (REFILL) {
        // (ring: the first unit has used all four fragments, their slot takes the block DB ahead; DMA before weights)
        if (XRING && t == 0) {
          dma_ring(kb_refill, bi % DB);
          __builtin_amdgcn_sched_barrier(0);
        }
        load_unit(t, kb_refill, u[t]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#ifdef W4A16_STAMPS
    STAMP(2 + min(n_done, 7)); ++n_done;
#endif
  };

  if (nblk > 0) {
    ex8 xa[4][MT], xb[4][MT];
    if constexpr (XLDS) {
      // activations first: they come from L2 and are back before the weights
      for (int bi = 0; bi < nblk; ++bi)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr)(smem_raw + xwave + bi * 1024), 16, xlane,
                                                 kb_of(bi) * 256, 0, 0);
    } else if constexpr (XRING) {
      dma_ring(kb0, 0);
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
      __builtin_amdgcn_sched_barrier(0);
      dma_ring(kb1, 1);
      __builtin_amdgcn_sched_barrier(0);
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
    // 5-32 rows: whole partial tiles (1 KiB per wave and tile), wave ti sums tile ti (and its up-half twin).  With
    // two row tiles a wave parks its partial tiles in its OWN ring slots where they fit (nobody else touches those, so
    // the ring and the parked tiles share LDS); else behind the ring, since other waves may still be reading theirs.
    constexpr int TMT = T * MT;                                      // partial tiles per wave: (t, mt) -> t MT + mt
    constexpr bool OWN = MT == 2 && TMT * 1024 <= DB * SLOT;
    constexpr int PSTRIDE = OWN ? DB * SLOT / 16 : TMT * 64;          // f32x4 between two waves' parked tiles
    f32x4* red = reinterpret_cast<f32x4*>(smem_raw + ((XRING && !OWN) ? W * DB * SLOT : 0));
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) red[w * PSTRIDE + (t * MT + mt) * 64 + lane] = acc[t][mt];
    __syncthreads();
    STAMP(11);
    if (w >= H * MT) return;
    const int ti = w / MT, rt = w % MT;  // column tile, row tile
    f32x4 v = red[w * 64 + lane], vu = v;
#pragma unroll
    for (int s = 1; s < W; ++s) v += red[s * PSTRIDE + w * 64 + lane];
    if constexpr (ACT) {
      vu = red[(ti + H) * 64 + lane];
#pragma unroll
      for (int s = 1; s < W; ++s) vu += red[s * PSTRIDE + (ti + H) * 64 + lane];
    }
    auto finish = [&](f32x4 r, int n_bias) {
      ex4 h = AT::round(r);
      if (a.bias != nullptr) h = AT::add(h, *reinterpret_cast<const ex4*>(reinterpret_cast<const elem*>(a.bias) + n_bias));
      return h;
    };
    const int nt = nt0 + ti, n = nt * 16 + kq * 4, tok = 16 * rt + n16;
    if (tok >= a.M) return;
    if constexpr (ACT) {
      const int NH = a.N / 2;
      if (nt < NTH) *reinterpret_cast<ex4*>(reinterpret_cast<elem*>(a.out) + (size_t)tok * NH + n) = silu_mul(finish(v, n), finish(vu, NH + n));
    } else {
      if (nt < a.NT && n < a.N) {
        ex4* dst = reinterpret_cast<ex4*>(reinterpret_cast<elem*>(a.out) + (size_t)tok * a.N + n);
        ex4 hv = finish(v, n);
        if (a.add_out) hv = AT::add(*dst, hv);
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
    for (int t = 0; t < T; ++t) red[((w * T + t) * 4 + n16) * 4 + kq] = acc[t][0];
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
    // What this rests on, beyond the language's memory model (relaxed accesses promise no order between
    // the stores and the ticket): on gfx950 a device-scope store (`global_store .. sc1`) goes through to memory
    // and leaves the wave's vmcnt only once it is acknowledged there, so after `s_waitcnt vmcnt(0)` this wave's
    // sums are visible to every XCD before its ticket is drawn; the last arriver's loads (`.. sc1`: served
    // past its CU's cache) are issued by the very wave whose returning atomic told it that it is last, after
    // the value is back; and the compiler keeps relaxed atomic loads behind the branch on that value.  Only
    // this one wave of a workgroup stores, signals and loads.
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
    ex4 h = AT::round(r);
    if (a.bias != nullptr) h = AT::add(h, *reinterpret_cast<const ex4*>(reinterpret_cast<const elem*>(a.bias) + n_bias));
    return h;
  };
  const int nt = nt0 + ti, n = nt * 16 + grp * 4;
  if constexpr (ACT) {
    const int NH = a.N / 2;
    if (nt < NTH) *reinterpret_cast<ex4*>(reinterpret_cast<elem*>(a.out) + (size_t)tok * NH + n) = silu_mul(finish(v, n), finish(vu, NH + n));
  } else {
    if (nt < a.NT && n < a.N) {
      ex4* dst = reinterpret_cast<ex4*>(reinterpret_cast<elem*>(a.out) + (size_t)tok * a.N + n);
      ex4 hv = finish(v, n);
      if (a.add_out) hv = AT::add(*dst, hv);
      *dst = hv;
    }
  }
  }  // pass
  STAMP(12);
#endif
}

template <int T, int W, int DB, class AT = ActF16>
int launch_decode(const GemmArgs& a, hipStream_t s, int batch, bool act, int splits, void* ws, size_t ws_bytes) {
  constexpr bool kPlain = T <= 8;
  constexpr bool kAct = T % 2 == 0 && T / 2 <= 8 && std::is_same<AT, ActF16>::value;  // (the fused activation: fp16 only)
  if (!decode_ok(a) || (act && a.N % 32)) return W4A16_E_SHAPE;
  if (act ? !kAct : !kPlain) return W4A16_E_SHAPE;  // (before any division by tiles per workgroup: T / 2 is 0 for T = 1)
  if (splits < 1 || splits > kMaxSplits) return W4A16_E_SHAPE;
  const bool xlds = a.M <= kDecodeMaxRows;
  if (!xlds && splits > 1) return W4A16_E_SHAPE;  // the K split is built for up to 4 rows
  if (splits > 1 && (act ? T / 2 : T) > 4) return W4A16_E_SHAPE;  // and up to 4 tiles or tile pairs
  const int kb_per_split = ceil_div(a.KB, splits);
  splits = ceil_div(a.KB, kb_per_split);  // no empty split
  const int nb_max = ceil_div(kb_per_split, W);
  // 5-16 rows: through the ring where DB slots of 4 KiB per wave fit beside the partial tiles, else fragments from L2
#ifndef W4A16_DECODE_RING
#define W4A16_DECODE_RING 1
#endif
  constexpr size_t ring_lds = (size_t)W * DB * 4096 + (size_t)W * T * 1024;
  const bool two = a.M > kDecodeMaxRowsL2;  // 17-32 rows: two row tiles through the ring, or not at all
  constexpr bool kTwo = decode_two_row_tiles_built(T, W, DB);
  if (two && (!kTwo || act || !W4A16_DECODE_RING)) return W4A16_E_SHAPE;
  const bool ring = W4A16_DECODE_RING && !xlds && !two && ring_lds <= 160 * 1024;
  const size_t lds = two ? decode_two_row_tiles_lds(T, W, DB)
                         : ring ? ring_lds : decode_lds_bytes(T, W, nb_max + 1, a.M, a.G);  // (+ 1: a wave may take a block from another)
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
  constexpr int kRingXm = ring_lds <= 160 * 1024 ? 2 : 0;  // (no ring instantiation for shapes it never fits)
  constexpr int kTwoXm = kTwo ? 3 : 0;
#define W4A16_PICK(SZS_)                                                                         \
  if (szs == SZS_) {                                                                             \
    if (act) {                                                                                   \
      if constexpr (kAct) kern = xlds ? gemm_decode_kernel<T, W, DB, SZS_, true, 1, AT>          \
                                 : ring ? gemm_decode_kernel<T, W, DB, SZS_, true, kRingXm, AT>  \
                                        : gemm_decode_kernel<T, W, DB, SZS_, true, 0, AT>;       \
    } else {                                                                                     \
      if constexpr (kPlain) kern = xlds ? gemm_decode_kernel<T, W, DB, SZS_, false, 1, AT>          \
                                   : two  ? gemm_decode_kernel<T, W, DB, SZS_, false, kTwoXm, AT>   \
                                   : ring ? gemm_decode_kernel<T, W, DB, SZS_, false, kRingXm, AT>  \
                                          : gemm_decode_kernel<T, W, DB, SZS_, false, 0, AT>;       \
    }                                                                                            \
  }
  W4A16_PICK(4) W4A16_PICK(2) W4A16_PICK(1)
#undef W4A16_PICK
  if (!kern) return W4A16_E_SHAPE;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      w4a16_note_hip_error((int)e);
      return W4A16_E_LAUNCH;
    }
  }
  hipLaunchKernelGGL(kern, dim3(gx, splits, batch), dim3(64 * W), lds, s, a, e);
  return launched();
}


}  // namespace

// `depth`: blocks a wave keeps in flight (1 or 2; 0 = the default for the shape)
int dispatch_decode(const GemmArgs& a, int t, int waves, hipStream_t s, int batch, bool act, int splits, void* ws,
                    size_t ws_bytes, int depth, bool bf16) {
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
  if (a.M > kDecodeMaxRowsL2) depth = 1;  // two row tiles: built with one block in flight
  if (depth == 0) {
    const int streams = t * waves;
    depth = (streams < 24 || (streams < 32 && a.K >= 6144) ||
             (streams == 32 && waves == 16 && (a.M > kDecodeMaxRows || a.K >= 12800))) ? 2 : 1;
    // 4 tiles x 4 waves (several workgroups to a CU): one block up to K = 6144 (4096 x 28672 at 1 row 14.0 us
    // against 14.9), two beyond (8 x 14336 x 4096: 46.6 against 52.2)
    if (t == 4 && waves == 4 && a.K < 6144) depth = 1;
  }
  if (bf16) {
#define X(T_, W_)                                                                                     \
  if (t == T_ && waves == W_)                                                                         \
    return depth == 1 ? launch_decode<T_, W_, 1, ActBf16>(a, s, batch, act, splits, ws, ws_bytes)     \
                      : launch_decode<T_, W_, 2, ActBf16>(a, s, batch, act, splits, ws, ws_bytes);
    W4A16_FOR_EACH_DECODE(X)
#undef X
    return W4A16_E_SHAPE;
  }
#define X(T_, W_)                                                                                     \
  if (t == T_ && waves == W_)                                                                         \
    return depth == 1 ? launch_decode<T_, W_, 1>(a, s, batch, act, splits, ws, ws_bytes)              \
                      : launch_decode<T_, W_, 2>(a, s, batch, act, splits, ws, ws_bytes);
  W4A16_FOR_EACH_DECODE(X)
#undef X
  return W4A16_E_SHAPE;
}

}  // namespace w4a16

W4A16_STAMPS_ACCESSOR
