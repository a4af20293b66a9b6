// Developer micro-benchmark: v_mfma_f32_32x32x16_f16 register loops with and without the int4 -> fp16
// conversion feeding them, and with the B fragments re-read from LDS (no global memory traffic).
// MT = 32-row tiles of activations per weight fragment, T = 32-column weight tiles per k-step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
__device__ __forceinline__ f16x2 as_h2(uint32_t v) { return __builtin_bit_cast(f16x2, v); }
__device__ __forceinline__ f16x8 dq(uint32_t w, f16x2 s, f16x2 nz1, f16x2 nz2, uint32_t lo, uint32_t hi, uint32_t mg) {
  const uint32_t t = w >> 8;
  const f16x2 c16 = f16x2{(f16)0.0625f, (f16)0.0625f};
  f16x2 d0 = (as_h2((w & lo) | mg) + nz1) * s;
  f16x2 d1 = (as_h2((w & hi) | mg) * c16 + nz2) * s;
  f16x2 d2 = (as_h2((t & lo) | mg) + nz1) * s;
  f16x2 d3 = (as_h2((t & hi) | mg) * c16 + nz2) * s;
  return f16x8{d0.x, d0.y, d1.x, d1.y, d2.x, d2.y, d3.x, d3.y};
}
// MODE 0: MFMA only.  1: T conversions per k-step feed the MFMAs.  2: as 1, B fragments read from LDS each k-step.
template <int MT, int T, int MODE, int LB>
__global__ void __launch_bounds__(LB) k(float* out, int iters, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[MT * 32 * 256];
  f32x16 acc[T][MT];
  for (int t = 0; t < T; ++t) for (int m = 0; m < MT; ++m) for (int i = 0; i < 16; ++i) acc[t][m][i] = 0.f;
  f16x8 xf[MT];
  for (int m = 0; m < MT; ++m) for (int i = 0; i < 8; ++i) xf[m][i] = (f16)(0.001f * ((threadIdx.x & 63) + m + i));
  if (MODE == 2) {
    for (int i = threadIdx.x; i < MT * 32 * 256 / 2; i += blockDim.x) reinterpret_cast<f16*>(lds)[i] = (f16)(0.001f * (i & 255));
    __syncthreads();
  }
  uint32_t lo = 0x000f000fu, hi = 0x00f000f0u, mg = 0x64006400u;
  asm volatile("" : "+s"(lo), "+s"(hi)); asm volatile("" : "+v"(mg));
  uint32_t w[T];
  for (int t = 0; t < T; ++t) w[t] = seed * (threadIdx.x + 17 * t + 1);
  f16x2 s = f16x2{(f16)0.01f, (f16)0.01f}, nz1 = f16x2{(f16)-1030.f, (f16)-1030.f}, nz2 = f16x2{(f16)-70.f, (f16)-70.f};
  f16x8 af[T];
  for (int t = 0; t < T; ++t) af[t] = dq(w[t], s, nz1, nz2, lo, hi, mg);
  const int lane = threadIdx.x & 63;
  const int row = lane & 31, kh = lane >> 5;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2) {
      // k-step j of a 128-k slab: 16-byte chunk 2 j + kh of the row, XOR-swizzled by the row
      const int j = it & 7;
#pragma unroll
      for (int m = 0; m < MT; ++m)
        xf[m] = *reinterpret_cast<const f16x8*>(lds + (m * 32 + row) * 256 + (((2 * j + kh) ^ (row & 15)) << 4));
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (MODE >= 1) { w[t] = w[t] * 1664525u + 1013904223u; af[t] = dq(w[t], s, nz1, nz2, lo, hi, mg); }
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t], xf[m], acc[t][m], 0, 0, 0);
    }
  }
  float r = 0.f;
  for (int t = 0; t < T; ++t) for (int m = 0; m < MT; ++m) for (int i = 0; i < 16; ++i) r += acc[t][m][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MT, int T, int MODE, int LB>
void run(const char* name, float* out, int iters) {
  const int blocks = 256, threads = LB;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MT, T, MODE, LB>), dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MT, T, MODE, LB>), dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  const double per_simd = (double)iters * T * MT * (threads / 64) / 4.0;
  const double mfmas = per_simd * 4 * blocks;
  const double tf = mfmas * 32768.0 / (ms * 1e-3) * 1e-12;
  printf("%-40s MT=%d T=%d waves/SIMD=%d: %8.3f ms %7.1f TFLOP/s  %.1f cyc@2.4GHz per MFMA per SIMD\n", name, MT, T,
         threads / 256, ms, tf, (ms * 1e-3 * 2.4e9) / per_simd);
}
int main() {
  float* out; CK(hipMalloc(&out, 1 << 22));
  const int it = 2000;
  run<2, 3, 0, 256>("mfma32 only", out, it);
  run<2, 3, 0, 512>("mfma32 only", out, it);
  run<2, 3, 1, 256>("convert + mfma32", out, it);
  run<2, 3, 1, 512>("convert + mfma32", out, it);
  run<2, 1, 1, 512>("convert + mfma32", out, it);
  run<2, 1, 1, 1024>("convert + mfma32", out, it);
  run<4, 1, 1, 512>("convert + mfma32", out, it);
  run<4, 3, 1, 256>("convert + mfma32", out, it);
  run<4, 3, 1, 512>("convert + mfma32", out, it);
  run<2, 3, 2, 256>("convert + mfma32, B from LDS", out, it);
  run<2, 3, 2, 512>("convert + mfma32, B from LDS", out, it);
  run<4, 1, 2, 512>("convert + mfma32, B from LDS", out, it);
  run<4, 3, 2, 512>("convert + mfma32, B from LDS", out, it);
  run<1, 3, 1, 512>("convert + mfma32", out, it);
  run<1, 3, 1, 1024>("convert + mfma32", out, it);
  return 0;
}
