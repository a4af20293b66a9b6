// Developer micro-benchmark: how fast can ONE short kernel stream ~23 MB from HBM on this box?
// Build: hipcc -O3 --offload-arch=gfx950 -o scripts/membench scripts/membench.hip
// Run under rocprofv3 --kernel-trace --stats and read the per-kernel durations.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

// Each thread reads PER 16-byte words, strided by the wave's footprint so every
// wave-instruction reads 1 KiB contiguous; all loads are issued before any use.
template <int PER, bool NT>
__global__ void read_kernel(const u32x4* __restrict__ p, uint32_t* __restrict__ out, size_t n16) {
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const size_t base = wave * (size_t)(64 * PER) + lane;
  u32x4 v[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    size_t idx = base + (size_t)i * 64;
    if (idx >= n16) idx = lane;
    v[i] = NT ? __builtin_nontemporal_load(p + idx) : p[idx];
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) s ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  if (s == 0x12345678u) out[wave] = s;  // practically never; keeps the loads alive
}

template <int PER, bool NT>
void run(const char* name, u32x4** bufs, int nbuf, uint32_t* out, size_t n16, int threads, int reps) {
  const size_t waves = (n16 + 64 * PER - 1) / (64 * PER);
  const size_t blocks = (waves * 64 + threads - 1) / threads;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int r = 0; r < 3; ++r)
    hipLaunchKernelGGL((read_kernel<PER, NT>), dim3(blocks), dim3(threads), 0, 0, bufs[r % nbuf], out, n16);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL((read_kernel<PER, NT>), dim3(blocks), dim3(threads), 0, 0, bufs[r % nbuf], out, n16);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  printf("%-28s blocks=%6zu threads=%4d : %7.2f us/launch (wall, back-to-back)  %7.1f GB/s\n", name, blocks,
         threads, us, n16 * 16.0 / us * 1e-3);
}

int main(int argc, char** argv) {
  const size_t bytes = argc > 1 ? (size_t)atoll(argv[1]) : 23425024;  // weight-side bytes of 4096x11008
  const int nbuf = 16;
  const size_t n16 = bytes / 16;
  u32x4* bufs[nbuf];
  for (int i = 0; i < nbuf; ++i) {
    CK(hipMalloc(&bufs[i], n16 * 16));
    CK(hipMemset(bufs[i], i + 1, n16 * 16));
  }
  uint32_t* out;
  CK(hipMalloc(&out, 1 << 20));
  CK(hipDeviceSynchronize());
  const int reps = 320;
  run<6, true>("per6 nt t1024", bufs, nbuf, out, n16, 1024, reps);
  run<6, false>("per6 plain t1024", bufs, nbuf, out, n16, 1024, reps);
  run<6, true>("per6 nt t512", bufs, nbuf, out, n16, 512, reps);
  run<6, true>("per6 nt t256", bufs, nbuf, out, n16, 256, reps);
  run<4, true>("per4 nt t256", bufs, nbuf, out, n16, 256, reps);
  run<4, false>("per4 plain t256", bufs, nbuf, out, n16, 256, reps);
  run<2, true>("per2 nt t256", bufs, nbuf, out, n16, 256, reps);
  run<1, true>("per1 nt t256", bufs, nbuf, out, n16, 256, reps);
  run<8, true>("per8 nt t256", bufs, nbuf, out, n16, 256, reps);
  run<8, true>("per8 nt t512", bufs, nbuf, out, n16, 512, reps);
  run<12, true>("per12 nt t512", bufs, nbuf, out, n16, 512, reps);
  run<16, true>("per16 nt t256", bufs, nbuf, out, n16, 256, reps);
  run<16, true>("per16 nt t1024", bufs, nbuf, out, n16, 1024, reps);
  // same buffer every time: Infinity-Cache resident
  u32x4* one[1] = {bufs[0]};
  run<6, true>("per6 nt t1024 cached", one, 1, out, n16, 1024, reps);
  run<4, false>("per4 plain t256 cached", one, 1, out, n16, 256, reps);
  return 0;
}
