// AWQ W4A16 GEMM for gfx950 (MI355X): the launch-shape choosers and the C ABI of libw4a16.so.
// The kernels live in one translation unit per family (w4a16_decode.hip, w4a16_streaming.hip, w4a16_tiled.hip,
// w4a16_wide.hip, w4a16_wide_shared.hip, w4a16_stock.hip); w4a16_common.h holds what they share.
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


#include "w4a16_common.h"

#ifdef W4A16_UNITY
// Diagnostic builds (-DW4A16_STAMPS -DW4A16_UNITY, this file alone): the whole library as one translation unit, so
// that the one stamp buffer of w4a16_common.h serves every kernel.
#include "w4a16_decode.hip"
#include "w4a16_stock.hip"
#include "w4a16_streaming.hip"
#include "w4a16_tiled.hip"
#include "w4a16_wide.hip"
#include "w4a16_wide_shared.hip"
#endif

namespace W4A16_HIDDEN w4a16 {
namespace {

thread_local int g_last_hip_error = 0;


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
  int ksplit = 1;   // private-slab 32x32x16 kernel: workgroups per output tile (K split through the workspace)
};

// Launch shape of the two 32x32x16 kernels: the built shape with the lowest estimated time,
//   ceil(workgroups / 256 CUs) x (K / 128 x a + b)   microseconds,
// a (per 128-k block) and b (prologue, reduction, stores) per shape, fitted by least squares on all seventeen
// shapes x 14 layer shapes (Llama-2-7B, Llama-2-70B as one of 8 ranks, Mixtral, 2/4/8-way shards of the
// benchmark layer) x 17-4096 rows (scripts/exp_shared_sweep.py), then each shape's pair scaled by a common factor
// where that lowers the loss of the CHOICE (scripts/fit_chooser.py --tune; whole rounds of 256
// undervalue the 4-wave 128 x 256 shape, two of which share a CU): mean loss against the best built shape 0.2 %
// on two sweeps of the current kernels (448 cases, every layer's weights rotated over 640 MB so that they come
// from HBM, not from the 256 MB memory-side cache); with each layer shape held out of fit and tuning in turn, 0.4 %
// (worst 29 %).  The 32-row x 32-column private-slab shape <1,1,8> is what narrow layers take up to 64 rows: 128
// column slices of a 4096-column layer x 2 row blocks fill the chip, where 64-column slices leave half of it idle
// (11008 x 4096 at 33 / 64 rows 12.1 / 13.6 us against 15.9 / 15.9; 8192 x 1280 at 17-128 rows 8.4-9.5 against 12.0).  The private-slab kernel splits K
// over the waves of a workgroup, so it keeps many workgroups on narrow layers at any row count (8192 x 1280 at
// 512 rows: 20 us against 58 for the best shared-slab shape); the shared-slab kernel wins where rows x columns
// alone fill the chip.
Choice choose_mfma32(int M, int K, int N, int batch, bool have_ws = false) {
  struct Shape {
    int mt, t, nw, kw;  // private slabs: nw = 1; shared slab: kw = 1
    double a, b;
  };
  static const Shape shapes[] = {
      {1, 1, 1, 8, 0.1245, 0.841}, {1, 2, 1, 8, 0.1751, 1.409}, {1, 3, 1, 4, 0.2302, 1.474}, {1, 3, 1, 8, 0.2046, 2.528},
      {2, 1, 1, 8, 0.2029, 2.183}, {2, 2, 1, 4, 0.2452, 2.968}, {2, 2, 1, 8, 0.2565, 3.395}, {2, 3, 1, 4, 0.2773, 3.642},
      {2, 3, 1, 8, 0.3107, 4.965}, {4, 3, 1, 4, 0.4608, 6.605}, {2, 1, 8, 1, 0.7621, 3.621}, {2, 2, 8, 1, 1.2463, 4.535},
      {2, 3, 8, 1, 1.7787, 5.336}, {4, 1, 8, 1, 1.1983, 3.983}, {4, 2, 4, 1, 1.0892, 3.330}, {4, 2, 8, 1, 2.0759, 5.227},
      {4, 3, 4, 1, 1.6615, 5.793}};
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
  // With a workspace, a layer of few columns and a long K (a tensor-parallel rank's qkv projection: 8192 x 1280) is
  // split over K as well: at 33-128 rows the unsplit shapes leave most CUs idle there.  Four workgroups per 64 x 64
  // tile meet through the workspace (see gemm_wide_kernel): 9.8 / 10.3 us at 64 / 128 rows against 12.3 / 12.4.
  // The exchange costs 5-7 us once the partial sums are large (two workgroups per 128 x 96 tile of the
  // 4096 x 11008 layer at 128 rows: 22.3 us against 17.9 unsplit), so nothing else is split.
  if (have_ws && M > 32 && M <= 128 && batch == 1 && K >= 8192) {
    const long long wgs = (long long)ceil_div(ceil_div(N, 32), 2) * ceil_div(M, 64);
    if (wgs <= 48) {
      best = Choice{true, 2, 2, 1, -4};
      best.ksplit = 4;
    }
  }
  return best;
}

Choice choose_cfg(int M, int K, int N, int batch = 1, bool act = false, bool have_ws = false) {
  const int NT = ceil_div(N, 16), KB = ceil_div(K, 128);
  // The 32x32x16 kernels read whole 128-k blocks through 32-bit byte offsets.  (With the fused activation a
  // tile is 16 gate + 16 up columns: as many tiles, and the same launch shapes, as the plain GEMM on N columns.)
  const bool wide_ok = K % 128 == 0 && (long long)NT * padded_kblocks(K) * 1024 < (1LL << 32) &&
                       (long long)M * K * 2 + kDmaImmBias < (1LL << 32);
  // From 33 rows: 10-40 % faster than the 16x16x32 kernels on every layer shape measured, and ahead of
  // dequantise + dense GEMM at every row count.
  if (wide_ok && M > 32) return choose_mfma32(M, K, N, batch, have_ws);
  // 17-32 rows of a single problem: 5-30 % over the 16x16x32 kernels from K = 2048 up (at 24 / 32 rows,
  // 4096 x 11008: 10.2 / 10.6 against 11.6 / 11.6 us; 4096 x 4096: 7.8 / 7.9 against 9.5 / 9.8; 4096 x 14336:
  // 11.0 / 11.4 against 14.6 / 14.8; 8192 x 1280: 12.0 against 15.4; 4096 x 1376: 7.7 / 7.8 against 9.2 / 10.6;
  // at 17 / 20 rows, 16 layer shapes: 4096 x 11008 9.3 / 9.0 against 11.2 / 11.7 for the streaming kernel, 4096 x 14336
  // 9.7 against 14.3 / 13.7, 11008 x 4096 15.3 against 16.9 / 18.7, 4096 x 22016 13.9 against 16.4; level within 4 %
  // at 17 rows on 4096 x 4096 and 4096 x 1376).  At K = 1024 the streaming kernel stays ahead.
  if (wide_ok && batch == 1 && M > kWideFromRows && K >= 2048) return choose_mfma32(M, K, N, batch);
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
  // 17-32 rows: the decode kernel with two 16-row tiles, in the launch shape it takes at 16 rows, where that shape is
  // built for two row tiles (plain GEMM on layers of up to 256 column tiles).
  if (M > kDecodeMaxRowsL2 && M <= kDecodeMaxRowsTwo && !act && K % 128 == 0) {
    Choice c16 = choose_with_decode(kDecodeMaxRowsL2, K, N, G, batch, act, false);
    if (c16.splits > 0 && decode_two_row_tiles_built(c16.t, c16.a, 1)) {
      c16.splits = 1;
      return c16;
    }
  }
  Choice c = choose_cfg(M, K, N, batch, act, have_ws);
  if (c.tiled || c.mt != 1 || M > kDecodeMaxRowsL2) return c;
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
}  // namespace w4a16

using namespace w4a16;

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

int w4a16_gemm_packed_f16_wide_split_cfg(const void* x, const void* packed_w, const void* packed_sz, const void* bias,
                                         void* out, int M, int K, int N, int G, int epilogue, int mt, int t, int kw,
                                         int splits, void* workspace, size_t workspace_bytes, void* stream) {
  if (epilogue != W4A16_EPI_NONE && epilogue != W4A16_EPI_SILU_MUL && epilogue != W4A16_EPI_ADD_TO_OUT)
    return W4A16_E_SHAPE;
  const bool act = epilogue == W4A16_EPI_SILU_MUL;
  if (act && N % 32) return W4A16_E_SHAPE;
  if (kw <= 0 || splits < 1 || splits > kMaxSplits) return W4A16_E_SHAPE;
  GemmArgs a;
  const int rc = fill_args(&a, x, packed_w, packed_sz, bias, out, M, K, N, G);
  if (rc) return rc;
  a.add_out = epilogue == W4A16_EPI_ADD_TO_OUT;
  clear_stale_error();
  return dispatch_wide(a, mt, t, kw, static_cast<hipStream_t>(stream), 1, act, splits, workspace, workspace_bytes);
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
    if (c.ksplit > 1) b_ = -(-c.b + 16 * (c.ksplit - 1));
  }
  if (mt == -1)  // decode kernel: (t, waves, splits + 256 * depth)
    return dispatch_decode(a, t, a_, s, batch, act, b_ & 255, ws, ws_bytes, b_ >> 8);
  if (b_ < 0) {  // 32x32x16 kernels: (mt, t, 1, -(kw + 16 (splits - 1))) private slabs, (mt, t, nw > 1, any b < 0) shared slab
    if (a_ > 1) return dispatch_wide_shared(a, mt, t, a_, s, batch, act);
    return dispatch_wide(a, mt, t, (-b_) & 15, s, batch, act, ((-b_) >> 4) + 1, ws, ws_bytes);
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

// bf16 activations: launch shape of the streaming kernel.  One 16-row tile up to 16 rows (16 waves over K where K is
// long enough to give every wave a block, up to three column tiles per workgroup), two row tiles beyond (8 waves,
// row blocks of 32 in gridDim.y, as many column tiles per workgroup as still leave a workgroup per CU).
constexpr int kBf16DecodeTag = -16;  // first entry of a bf16 launch shape that names the decode kernel
constexpr int kBf16SharedTag = -36;  // ... the shared-slab 32x32x16 kernel (four row tiles)
static void choose_bf16(int M, int K, int N, int* mt, int* t, int* waves) {
  const int NT = ceil_div(N, 16), KB = ceil_div(K, 128);
  if (M <= 16) {
    *mt = 1;
    if (KB >= 16 && NT <= 3 * 256) {
      *waves = 16;
      *t = NT <= 256 ? 1 : NT <= 512 ? 2 : 3;
    } else {
      *waves = 8;
      *t = NT <= 2 * 256 ? 2 : 4;
    }
    return;
  }
  *mt = 2;
  *waves = 8;
  const long long per_cu = (long long)NT * ceil_div(M, 32) / 256;
  *t = per_cu >= 4 ? 4 : per_cu >= 3 ? 3 : per_cu >= 2 ? 2 : 1;
}

int w4a16_gemm_packed_bf16(const void* x, const void* packed_w, const void* packed_sz, const void* bias, void* out,
                           int M, int K, int N, int G, int epilogue, int mt, int t, int waves, void* stream) {
  if (epilogue != W4A16_EPI_NONE && epilogue != W4A16_EPI_ADD_TO_OUT) return W4A16_E_SHAPE;
  GemmArgs a;
  const int rc = fill_args(&a, x, packed_w, packed_sz, bias, out, M, K, N, G);
  if (rc) return rc;
  a.add_out = epilogue == W4A16_EPI_ADD_TO_OUT;
  clear_stale_error();
  if (mt == 0) {
    // From 17 rows the 32x32x16 kernel with private slabs (whole 128-k blocks only): one 32-row tile up to 47 rows,
    // two beyond.
    const bool wide_ok = K % 128 == 0 && (long long)a.NT * a.KBS * 1024 < (1LL << 32) &&
                         (long long)M * K * 2 + kDmaImmBias < (1LL << 32);
    if (M > 16 && wide_ok) {
      // Beyond 256 rows the shared-slab kernel where the fp16 chooser takes it: its shape if that is built for bf16,
      // else the built one nearest in rows per launch.
      const Choice c = choose_mfma32(M, K, N, 1);
      if (M > 256 && K >= 256 && c.a > 1) {
        int st = c.t, snw = c.a;
        if (!(c.mt == 4 && wide_shared_bf16_built(4, st, snw))) st = M >= 768 && M < 2048 ? 3 : 2, snw = M >= 2048 ? 8 : 4;
        return dispatch_wide_shared(a, 4, st, snw, static_cast<hipStream_t>(stream), 1, false, true);
      }
      // the fp16 chooser's private-slab shape where that is built for bf16 (128 x 96 tiles at 129-256 rows of wide layers)
      if (c.a == 1 && c.b < 0 && wide_bf16_built(c.mt, c.t, -c.b))
        return dispatch_wide(a, c.mt, c.t, -c.b, static_cast<hipStream_t>(stream), 1, false, 1, nullptr, 0, true);
      const bool two = M >= 48;
      return dispatch_wide(a, two ? 2 : 1, two ? 3 : 2, two ? 4 : 8, static_cast<hipStream_t>(stream), 1, false, 1,
                           nullptr, 0, true);
    }
    // Up to 16 rows the decode kernel in the launch shape it takes for fp16 (no K split: no workspace here).
    if (M <= kDecodeMaxRowsTwo) {
      const Choice c = choose_with_decode(M, K, N, G, 1, false, false);
      if (c.splits > 0) return dispatch_decode(a, c.t, c.a, static_cast<hipStream_t>(stream), 1, false, 1, nullptr, 0, 0, true);
    }
    choose_bf16(M, K, N, &mt, &t, &waves);
  }
  if (mt < 0) {
    if (mt == kBf16DecodeTag)  // (tag, t, waves): the decode kernel, up to 16 rows
      return dispatch_decode(a, t, waves, static_cast<hipStream_t>(stream), 1, false, 1, nullptr, 0, 0, true);
    if (mt == kBf16SharedTag)  // (tag, t, nw): the shared-slab 32x32x16 kernel with four row tiles
      return dispatch_wide_shared(a, 4, t, waves, static_cast<hipStream_t>(stream), 1, false, true);
    // (-mt, t, waves as kw): the 32x32x16 kernel, as w4a16_gemm_packed_f16_wide_cfg names its shapes
    return dispatch_wide(a, -mt, t, waves, static_cast<hipStream_t>(stream), 1, false, 1, nullptr, 0, true);
  }
  return dispatch_packed_bf16(a, mt, t, waves, static_cast<hipStream_t>(stream));
}

int w4a16_built_shape_bf16(int i, int* shape) {
  if (!shape || i < 0) return W4A16_E_SHAPE;
  static const int built[][3] = {
#define X(MT_, T_, W_) {MT_, T_, W_},
      W4A16_FOR_EACH_BF16(X)
#undef X
      {-2, 3, 4}, {-1, 2, 8}, {-4, 3, 4}, {-1, 3, 4}, {-2, 2, 4}, {-1, 1, 8},  // the private-slab 32x32x16 kernel (wide_bf16_built)
#define X(T_, W_) {kBf16DecodeTag, T_, W_},
      W4A16_FOR_EACH_DECODE(X)  // the decode kernel
#undef X
      {kBf16SharedTag, 2, 4}, {kBf16SharedTag, 3, 4}, {kBf16SharedTag, 2, 8},  // the shared-slab kernel (wide_shared_bf16_built)
  };
  if (i >= (int)(sizeof(built) / sizeof(built[0]))) return W4A16_E_SHAPE;
  for (int j = 0; j < 3; ++j) shape[j] = built[i][j];
  return W4A16_OK;
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

size_t w4a16_wide_workspace_bytes(int M, int N, int batch, int epilogue, int mt, int t, int splits) {
  if (M <= 0 || N <= 0 || batch <= 0 || mt <= 0 || t <= 0 || splits <= 1 || splits > kMaxSplits) return 0;
  const long long tiles = (long long)batch * ceil_div(ceil_div(N, 32), t) * ceil_div(M, 32 * mt);
  return wide_ws_bytes(tiles, splits, mt, t, epilogue == W4A16_EPI_SILU_MUL);
}

size_t w4a16_workspace_bytes(int M, int K, int N, int G, int batch, int epilogue) {
  if (M <= 0 || K <= 0 || N <= 0 || batch <= 0 || check_common(K, N, G)) return 0;
  const bool act = epilogue == W4A16_EPI_SILU_MUL;
  const Choice c = choose_with_decode(M, K, N, G, batch, act, true);
  if (c.ksplit > 1) return w4a16_wide_workspace_bytes(M, N, batch, epilogue, c.mt, c.t, c.ksplit);
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
  if (c.ksplit > 1) {
    shape[0] = W4A16_KERNEL_MFMA32, shape[1] = c.mt, shape[2] = c.t, shape[3] = c.ksplit, shape[4] = -c.b;
    return W4A16_OK;
  }
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

}  // extern "C"
