// libw4a16.so: the LDS-staged GEMM kernel (v_mfma_f32_16x16x32_f16, activation slabs through swizzled LDS)
// and its launcher.  Layout, arithmetic and the kernel families are described at the top of w4a16.hip.

#include "w4a16_common.h"

namespace W4A16_HIDDEN w4a16 {
namespace {

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
      w4a16_note_hip_error((int)e);
      return W4A16_E_LAUNCH;
    }
  }
  // panels of a few row blocks once there are more than 8 of them (see map_workgroup)
  const int panel_rows = m_blocks > 8 ? kPanelRows : m_blocks;
  GemmArgs am = a;
  am.mb_magic = row_block_magic((long long)n_slices * m_blocks, m_blocks, panel_rows);
  hipLaunchKernelGGL(kern, dim3(blocks, 1, batch), dim3(64 * NW * KW), lds, s, n_slices, m_blocks, panel_rows, am);
  return launched();
}


}  // namespace

int dispatch_tiled(const GemmArgs& a, int mt, int t, int nw, int kw, hipStream_t s, int batch,
                   bool act) {
#define X(MT_, T_, NW_, KW_) \
  if (mt == MT_ && t == T_ && nw == NW_ && kw == KW_) return launch_tiled<MT_, T_, NW_, KW_>(a, s, batch, act);
  W4A16_FOR_EACH_TILED(X)
#undef X
  return W4A16_E_SHAPE;
}


}  // namespace w4a16

W4A16_STAMPS_ACCESSOR
