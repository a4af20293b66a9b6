// Developer micro-benchmark: issue cost of the instructions the int4 -> fp16 conversion is made of, per
// wave and per SIMD, at 1, 2 and 4 waves per SIMD (s_memtime around an unrolled loop of independent
// instructions; one workgroup per CU).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

// KIND: 0 v_pk_mul_f16, 1 v_pk_add_f16, 2 v_pk_fma_f16, 3 v_and_or_b32, 4 v_lshrrev_b32, 5 v_fma_f32,
//       6 v_pk_mul_f16 with an SGPR operand + op_sel, 7 v_fma_mixlo_f16, 8 v_mul_f16 (not packed),
//       9 whole conversion of a word (13 instructions) + one 16x16x32 MFMA, 10 the same without MFMA
template <int KIND>
__global__ void __launch_bounds__(1024) k(unsigned long long* stamps, float* sink, int iters) {
  uint32_t a0 = threadIdx.x * 0x01010101u + 0x3c003c00u, a1 = 0x3c003c00u, a2 = 0x38003800u, a3 = 0x3c013c01u;
  uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
  uint32_t lo = 0x000f000fu, hi = 0x00f000f0u, mg = 0x64006400u, s16 = 0x2c002c00u;
  asm volatile("" : "+s"(lo), "+s"(hi), "+s"(s16));
  asm volatile("" : "+v"(mg), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  f16x8 xf;
  for (int i = 0; i < 8; ++i) xf[i] = (f16)(0.01f * (threadIdx.x & 63));
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0) {
      REP32(asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(r0) : "v"(a0), "v"(a1));)
    } else if constexpr (KIND == 1) {
      REP32(asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(r0) : "v"(a0), "v"(a1));)
    } else if constexpr (KIND == 2) {
      REP32(asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(r0) : "v"(a0), "v"(a1), "v"(a2));)
    } else if constexpr (KIND == 3) {
      REP32(asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(r0) : "v"(a0), "s"(lo), "v"(mg));)
    } else if constexpr (KIND == 4) {
      REP32(asm volatile("v_lshrrev_b32 %0, 8, %1" : "=v"(r0) : "v"(a0));)
    } else if constexpr (KIND == 5) {
      REP32(asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(a0), "v"(a1), "v"(a2));)
    } else if constexpr (KIND == 6) {
      REP32(asm volatile("v_pk_fma_f16 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r0) : "v"(a0), "s"(s16), "v"(a2));)
    } else if constexpr (KIND == 7) {
      REP32(asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "+v"(r0) : "v"(a0), "v"(a1), "v"(a2));)
    } else if constexpr (KIND == 8) {
      REP32(asm volatile("v_mul_f16 %0, %1, %2" : "=v"(r0) : "v"(a0), "v"(a1));)
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile(
            "v_lshrrev_b32 %0, 8, %8\n"
            "v_and_or_b32 %1, %8, %9, %11\n"
            "v_and_or_b32 %2, %8, %10, %11\n"
            "v_and_or_b32 %3, %0, %9, %11\n"
            "v_and_or_b32 %4, %0, %10, %11\n"
            "v_pk_add_f16 %1, %1, %12\n"
            "v_pk_fma_f16 %2, %2, %13, %14 op_sel_hi:[1,0,1]\n"
            "v_pk_add_f16 %3, %3, %12\n"
            "v_pk_fma_f16 %4, %4, %13, %14 op_sel_hi:[1,0,1]\n"
            "v_pk_mul_f16 %1, %1, %15\n"
            "v_pk_mul_f16 %2, %2, %15\n"
            "v_pk_mul_f16 %3, %3, %15\n"
            "v_pk_mul_f16 %4, %4, %15\n"
            : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
            : "v"(a0), "s"(lo), "s"(hi), "v"(mg), "v"(a1), "s"(s16), "v"(a2), "v"(a3));
        if constexpr (KIND == 9) {
          f16x8 af = __builtin_bit_cast(f16x8, (uint4){r1, r2, r3, r4});
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, xf, acc, 0, 0, 0);
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 16 + wave] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + (float)(r0 + r1 + r2 + r3 + r4);
}

template <int KIND>
void run(const char* name, unsigned long long* stamps, float* sink, int per_iter) {
  const int iters = 200;
  for (int threads : {256, 512, 1024}) {
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(threads), 0, 0, stamps, sink, iters);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(threads), 0, 0, stamps, sink, iters);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[256 * 16];
    CK(hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost));
    double sum = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) { sum += (double)h[b * 16 + w]; ++n; }
    const double cyc_wave = sum / n / ((double)iters * per_iter);
    const int wps = threads / 256;
    printf("%-44s waves/SIMD=%d: %6.2f cyc per instruction per wave, %6.2f per SIMD  (%.3f ms)\n", name, wps, cyc_wave,
           cyc_wave / wps, ms);
  }
}

int main() {
  unsigned long long* stamps; float* sink;
  CK(hipMalloc(&stamps, 256 * 16 * 8)); CK(hipMalloc(&sink, 256 * 1024 * 4));
  run<5>("v_fma_f32", stamps, sink, 32);
  run<0>("v_pk_mul_f16", stamps, sink, 32);
  run<1>("v_pk_add_f16", stamps, sink, 32);
  run<2>("v_pk_fma_f16", stamps, sink, 32);
  run<6>("v_pk_fma_f16 sgpr op_sel", stamps, sink, 32);
  run<3>("v_and_or_b32", stamps, sink, 32);
  run<4>("v_lshrrev_b32", stamps, sink, 32);
  run<7>("v_fma_mixlo_f16", stamps, sink, 32);
  run<8>("v_mul_f16", stamps, sink, 32);
  run<10>("conversion of a word (13 instr), per word", stamps, sink, 8);
  run<9>("conversion + mfma16x16x32, per word", stamps, sink, 8);
  return 0;
}
