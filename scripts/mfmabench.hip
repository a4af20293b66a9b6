// Developer micro-benchmark: MFMA-only and dequant+MFMA register loops (no memory traffic).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
__device__ __forceinline__ f16x2 as_h2(uint32_t v) { return __builtin_bit_cast(f16x2, v); }
__device__ uint32_t g_lo, g_hi, g_magic;
__device__ __forceinline__ f16x8 dq(uint32_t w, f16x2 s, f16x2 nz1, f16x2 nz2) {
  const uint32_t t = w >> 8;
  const f16x2 c16 = f16x2{(f16)0.0625f, (f16)0.0625f};
  uint32_t lo = 0x000f000fu, hi = 0x00f000f0u, mg = 0x64006400u;
  asm volatile("" : "+s"(lo), "+s"(hi)); asm volatile("" : "+v"(mg));
  f16x2 d0 = (as_h2((w & lo) | mg) + nz1) * s;
  f16x2 d1 = (as_h2((w & hi) | mg) * c16 + nz2) * s;
  f16x2 d2 = (as_h2((t & lo) | mg) + nz1) * s;
  f16x2 d3 = (as_h2((t & hi) | mg) * c16 + nz2) * s;
  return f16x8{d0.x, d0.y, d1.x, d1.y, d2.x, d2.y, d3.x, d3.y};
}
// MODE 0: MFMA only. MODE 1: T dequants per step feeding the MFMAs.
template <int MT, int T, int MODE, int LB>
__global__ void __launch_bounds__(LB) k(float* out, int iters, uint32_t seed) {
  f32x4 acc[T][MT];
  for (int t = 0; t < T; ++t) for (int m = 0; m < MT; ++m) acc[t][m] = f32x4{0, 0, 0, 0};
  f16x8 xf[MT];
  for (int m = 0; m < MT; ++m) for (int i = 0; i < 8; ++i) xf[m][i] = (f16)(0.001f * (threadIdx.x + m + i));
  uint32_t w[T];
  for (int t = 0; t < T; ++t) w[t] = seed * (threadIdx.x + 17 * t + 1);
  f16x2 s = f16x2{(f16)0.01f, (f16)0.01f}, nz1 = f16x2{(f16)-1030.f, (f16)-1030.f}, nz2 = f16x2{(f16)-70.f, (f16)-70.f};
  f16x8 af[T];
  for (int t = 0; t < T; ++t) af[t] = dq(w[t], s, nz1, nz2);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (MODE == 1) { w[t] = w[t] * 1664525u + 1013904223u; af[t] = dq(w[t], s, nz1, nz2); }
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[t], xf[m], acc[t][m], 0, 0, 0);
    }
  }
  f32x4 r = f32x4{0, 0, 0, 0};
  for (int t = 0; t < T; ++t) for (int m = 0; m < MT; ++m) r += acc[t][m];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r.x + r.y + r.z + r.w;
}
template <int MT, int T, int MODE, int LB = 512>
void run(const char* name, float* out, int blocks, int threads, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MT, T, MODE, LB>), dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MT, T, MODE, LB>), dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  double mfmas = (double)blocks * (threads / 64) * iters * T * MT;
  double tf = mfmas * 16384.0 / (ms * 1e-3) * 1e-12;
  double cyc_per_mfma = (ms * 1e-3 * 2.4e9) / ((double)iters * T * MT * (threads / 64) * blocks / (256.0 * 4));
  printf("%-36s blocks=%d thr=%d: %8.3f ms  %7.1f TFLOP/s  %.1f cyc@2.4GHz per MFMA per SIMD\n", name, blocks, threads, ms, tf, cyc_per_mfma);
}
int main() {
  float* out; CK(hipMalloc(&out, 1 << 22));
  const int it = 2000;
  run<8, 3, 0>("mfma only MT8 T3 1w/simd", out, 256, 256, it);
  run<8, 3, 0>("mfma only MT8 T3 2w/simd", out, 256, 512, it);
  run<8, 3, 1>("dequant+mfma MT8 T3 1w/simd", out, 256, 256, it);
  run<8, 3, 1>("dequant+mfma MT8 T3 2w/simd", out, 256, 512, it);
  run<4, 6, 1>("dequant+mfma MT4 T6 2w/simd", out, 256, 512, it);
  run<2, 6, 1>("dequant+mfma MT2 T6 2w/simd", out, 256, 512, it);
  run<1, 3, 1>("dequant+mfma MT1 T3 2w/simd", out, 256, 512, it);
  run<1, 3, 1, 1024>("dequant+mfma MT1 T3 4w/simd", out, 256, 1024, it);
  run<1, 3, 0, 1024>("mfma only MT1 T3 4w/simd", out, 256, 1024, it);
  run<1, 3, 1, 256>("dequant+mfma MT1 T3 1w/simd", out, 256, 256, it);
  return 0;
}
