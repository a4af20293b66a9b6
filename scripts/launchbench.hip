// Developer micro-benchmark: cost of back-to-back (graph-replayed) launches of trivial kernels
// at several geometries: the floor under any ~5 us kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
extern __shared__ unsigned char smem[];
__global__ void empty_kernel(uint32_t* out, int touch_lds) {
  if (touch_lds) { smem[threadIdx.x] = 1; __syncthreads(); if (smem[(threadIdx.x + 1) % blockDim.x] == 7) out[0] = 1; }
  if (threadIdx.x == 0 && blockIdx.x == 0x7fffffff) out[0] = 1;
}
void run(const char* name, uint32_t* out, int blocks, int threads, size_t lds, int touch) {
  hipStream_t s; CK(hipStreamCreate(&s));
  if (lds > 65536) CK(hipFuncSetAttribute((const void*)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < 16; ++i) hipLaunchKernelGGL(empty_kernel, dim3(blocks), dim3(threads), lds, s, out, touch);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s));
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-40s blocks=%5d thr=%4d lds=%6zu: %6.2f us per launch\n", name, blocks, threads, lds, ms * 1e3 / 320);
}
int main() {
  uint32_t* out; CK(hipMalloc(&out, 4096));
  run("empty", out, 230, 1024, 0, 0);
  run("empty lds48k", out, 230, 1024, 49152, 0);
  run("empty lds48k touch", out, 230, 1024, 49152, 1);
  run("empty 512thr lds96k", out, 230, 512, 98304, 0);
  run("empty 512thr", out, 230, 512, 0, 0);
  run("empty 256thr x920", out, 920, 256, 0, 0);
  run("empty 256thr x256", out, 256, 256, 0, 0);
  run("empty 64thr x3680", out, 3680, 64, 0, 0);
  run("empty 1 block", out, 1, 64, 0, 0);
  return 0;
}
