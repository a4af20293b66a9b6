// Developer micro-benchmark: how fast does ONE CU take in an L2-resident activation matrix, the way the 32x32x16
// GEMM kernels fetch their slabs?  One 4-wave workgroup per CU, every wave walks its share of the 128-k blocks of a
// [rows][K] fp16 matrix (1 MiB at 128 x 4096) and fetches a [64 rows] x [128 k] slab per block as 16 instructions
// of 1 KiB.  Variants: LDS-DMA or register loads; slab rows as they lie in memory (8 rows x 128 bytes per
// instruction, rows K * 2 bytes apart) or a pre-arranged copy where an instruction reads 1 KiB contiguous; every
// workgroup walking K in the same order or each starting at its own block; 8 or 16 instructions in flight per wave.
// Build: hipcc -O3 --offload-arch=gfx950 -o scripts/xfeedbench scripts/xfeedbench.hip ; prints bytes per clock per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MODE bit 0: contiguous pieces (x is the pre-arranged copy); bit 1: register loads instead of LDS-DMA
// DEPTH: half slabs (8 instructions each) kept in flight beyond the one being issued (1 or 2... ring of DEPTH + 1)
template <int MODE, int DEPTH>
__global__ void __launch_bounds__(256) feed(const uint16_t* x, int K, int passes, int spread, unsigned long long* out,
                                            uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool CONTIG = MODE & 1, REGS = (MODE & 2) != 0;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int KB = K / 128;
  const int m0 = (blockIdx.x & 1) * 64;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x), 0, 128 * K * 2, 0x00020000);
  uint32_t off[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    if (CONTIG)  // pre-arranged: [row half][block][half slab][piece][1 KiB]
      off[p] = (uint32_t)(m0 / 64) * (KB * 16384) + p * 1024 + lane * 16;
    else
      off[p] = ((uint32_t)(m0 + 8 * p + (lane >> 3)) * K + (lane & 7) * 8) * 2u;
  }
  const int rot = spread ? (int)((blockIdx.x >> 1) * 5 % KB) : 0;
  constexpr int RING = DEPTH + 1;
  const int wbase = w * RING * 8192;
  u32x4 acc = {0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int slot = 0;
  for (int pass = 0; pass < passes; ++pass) {
    for (int kb0 = w; kb0 < KB; kb0 += 4) {
      int kb = kb0 + rot;
      if (kb >= KB) kb -= KB;
#pragma unroll
      for (int hp = 0; hp < 2; ++hp) {
        const uint32_t soff = CONTIG ? (uint32_t)(kb * 16384 + hp * 8192) : (uint32_t)((kb * 128 + hp * 64) * 2);
        if (REGS) {
          u32x4 v[8];
#pragma unroll
          for (int p = 0; p < 8; ++p) v[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off[p], soff, 0));
#pragma unroll
          for (int p = 0; p < 8; ++p) acc ^= v[p];
        } else {
#pragma unroll
          for (int p = 0; p < 8; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + wbase + slot * 8192 + p * 1024), 16, off[p], soff, 0, 0);
          wait_vmcnt<8 * DEPTH>();
          slot = slot + 1 == RING ? 0 : slot + 1;
        }
      }
    }
  }
  wait_vmcnt<0>();
  if (!REGS) acc = *reinterpret_cast<const u32x4*>(smem + wbase + lane * 16);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 4 + w] = t1 - t0;
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345679u) sink[0] = acc.x;
}

template <int MODE, int DEPTH>
void run(const char* name, const uint16_t* x, int K, int blocks, int spread, unsigned long long* out, uint32_t* sink) {
  const int passes = 20;
  const int lds = 4 * (DEPTH + 1) * 8192;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(feed<MODE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((feed<MODE, DEPTH>), dim3(blocks), dim3(256), lds, 0, x, K, passes, spread, out, sink);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(blocks * 4);
  CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> per_cu;
  for (int b = 0; b < blocks; ++b) {
    const unsigned long long mx = std::max(std::max(h[4 * b], h[4 * b + 1]), std::max(h[4 * b + 2], h[4 * b + 3]));
    per_cu.push_back((double)passes * 64 * K * 2 / (double)mx);
  }
  std::sort(per_cu.begin(), per_cu.end());
  printf("%-44s wgs %3d  bytes/clk/CU: min %5.1f  median %5.1f  max %5.1f\n", name, blocks, per_cu.front(),
         per_cu[per_cu.size() / 2], per_cu.back());
}

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 4096;
  uint16_t* x;
  CK(hipMalloc(&x, (size_t)128 * K * 2));
  CK(hipMemset(x, 0x3c, (size_t)128 * K * 2));
  unsigned long long* out;
  CK(hipMalloc(&out, 256 * 4 * 8));
  uint32_t* sink;
  CK(hipMalloc(&sink, 64));
  for (int blocks : {230, 256, 32}) {
    run<0, 1>("LDS-DMA rows, same order, 8 in flight", x, K, blocks, 0, out, sink);
    run<0, 2>("LDS-DMA rows, same order, 16 in flight", x, K, blocks, 0, out, sink);
    run<0, 3>("LDS-DMA rows, same order, 24 in flight", x, K, blocks, 0, out, sink);
    run<0, 2>("LDS-DMA rows, spread order, 16 in flight", x, K, blocks, 1, out, sink);
    run<1, 2>("LDS-DMA contiguous, same order, 16 in flight", x, K, blocks, 0, out, sink);
    run<1, 2>("LDS-DMA contiguous, spread, 16 in flight", x, K, blocks, 1, out, sink);
    run<2, 2>("register loads rows, same order", x, K, blocks, 0, out, sink);
    run<2, 2>("register loads rows, spread", x, K, blocks, 1, out, sink);
    run<3, 2>("register loads contiguous, same order", x, K, blocks, 0, out, sink);
    run<3, 2>("register loads contiguous, spread", x, K, blocks, 1, out, sink);
  }
  return 0;
}
