/*
 * CPU restatement of the reference's AWQ arithmetic in plain C.
 *
 * TEST INFRASTRUCTURE ONLY: loaded by tests/ and by bench.py's cpu_baseline leg,
 * never by the shipped package.  PARITY UNPINNED: the reference has no AWQ test
 * vectors and its CUDA/PTX kernel cannot be compiled in this image, so this file
 * (like oracle/awq_oracle.py, written independently of it) restates the
 * reference's sources:
 *   nibble order            csrc/quantization/awq/dequantize.cuh:17-84
 *   w = fp16((q - z) * s)   csrc/quantization/awq/gemm_kernels.cu:128-130,147-160
 *   dequantize layout       csrc/quantization/awq/gemm_kernels.cu:293-326
 *   split-K, fp16 slices    csrc/quantization/awq/gemm_kernels.cu:112-115,262-270,445
 *
 * fp16 is carried as uint16_t bit patterns; conversions are done by hand
 * (round-to-nearest-even), so no compiler fp16 support is needed.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const int kOrder[8] = {0, 4, 1, 5, 2, 6, 3, 7};

static float h2f(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else { /* subnormal: normalise */
      int e = -1;
      do {
        ++e;
        man <<= 1;
      } while (!(man & 0x400u));
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 112u) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

static uint16_t f2h(float f) { /* round to nearest even, like __float2half_rn / cvt.rn.f16.f32 */
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
  x &= 0x7fffffffu;
  if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((x > 0x7f800000u) ? 0x200u : 0)); /* inf / nan */
  if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); /* rounds to >= 65520 -> inf */
  if (x < 0x33000001u) return sign;                         /* < 2^-25 (or == 2^-25: ties to even 0) */
  int32_t exp = (int32_t)(x >> 23) - 127;
  uint32_t man = (x & 0x7fffffu) | 0x800000u;
  int shift;
  uint32_t hexp;
  if (exp < -14) { /* subnormal half */
    shift = 13 + (-14 - exp);
    hexp = 0;
  } else {
    shift = 13;
    hexp = (uint32_t)(exp + 15);
  }
  uint32_t q = man >> shift;
  const uint32_t rem = man & ((1u << shift) - 1u);
  const uint32_t halfway = 1u << (shift - 1);
  if (rem > halfway || (rem == halfway && (q & 1u))) ++q;
  /* q includes the implicit bit for normals; adding (hexp-1)<<10 folds a mantissa carry into the exponent */
  uint32_t out = (hexp == 0) ? q : (((hexp - 1u) << 10) + q);
  return (uint16_t)(sign | out);
}

static inline int nibble(uint32_t word, int j) { return (int)((word >> (4 * kOrder[j])) & 0xfu); }

/* out[K][N] fp16 bits.  qweight [K][N/8], scales [K/G][N] fp16 bits, qzeros [K/G][N/8]. */
void awq_c_dequantize(const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros, uint16_t* out, int K,
                      int N, int G) {
  const int NC = N / 8;
  for (int k = 0; k < K; ++k) {
    const int g = k / G;
    for (int c = 0; c < NC; ++c) {
      const uint32_t qw = (uint32_t)qweight[(size_t)k * NC + c];
      const uint32_t zw = (uint32_t)qzeros[(size_t)g * NC + c];
      for (int j = 0; j < 8; ++j) {
        const float d = (float)(nibble(qw, j) - nibble(zw, j)); /* exact, as sub.f16x2 is */
        const float s = h2f(scales[(size_t)g * N + 8 * c + j]);
        out[(size_t)k * N + 8 * c + j] = f2h(d * s); /* product exact in fp32; one rounding */
      }
    }
  }
}

/* out[M][N] fp16 bits.  faithful != 0: S-way interleaved split-K with fp16 slices (reference);
 * faithful == 0: one fp32 accumulation over all of K, rounded once.  Returns 0, or -1 on OOM. */
int awq_c_gemm(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
               uint16_t* out, int M, int K, int N, int G, int split_k, int faithful) {
  uint16_t* wh = (uint16_t*)malloc((size_t)K * N * sizeof(uint16_t));
  float* wf = (float*)malloc((size_t)K * N * sizeof(float));
  float* part = (float*)malloc((size_t)N * sizeof(float));
  float* tot = (float*)malloc((size_t)N * sizeof(float));
  if (!wh || !wf || !part || !tot) {
    free(wh); free(wf); free(part); free(tot);
    return -1;
  }
  awq_c_dequantize(qweight, scales, qzeros, wh, K, N, G);
  for (size_t i = 0; i < (size_t)K * N; ++i) wf[i] = h2f(wh[i]);
  const int S = faithful ? split_k : 1;
  for (int m = 0; m < M; ++m) {
    for (int n = 0; n < N; ++n) tot[n] = 0.f;
    for (int z = 0; z < S; ++z) {
      for (int n = 0; n < N; ++n) part[n] = 0.f;
      for (int k = 0; k < K; ++k) {
        if (faithful && ((k / 32) % S) != z) continue;
        const float xv = h2f(x[(size_t)m * K + k]);
        const float* wr = wf + (size_t)k * N;
        for (int n = 0; n < N; ++n) part[n] += xv * wr[n];
      }
      if (faithful) {
        for (int n = 0; n < N; ++n) tot[n] += h2f(f2h(part[n])); /* slice stored as fp16, summed in fp32 */
      } else {
        for (int n = 0; n < N; ++n) tot[n] = part[n];
      }
    }
    for (int n = 0; n < N; ++n) out[(size_t)m * N + n] = f2h(tot[n]);
  }
  free(wh); free(wf); free(part); free(tot);
  return 0;
}

/* helpers exported so the tests can check the hand-written conversions against numpy */
uint16_t awq_c_f2h(float f) { return f2h(f); }
float awq_c_h2f(uint16_t h) { return h2f(h); }
