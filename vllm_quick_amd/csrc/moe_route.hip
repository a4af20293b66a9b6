// Expert routing and the weighted sum over experts of a Mixtral-style block, one launch each.
//
// The reference's block (vllm/model_executor/models/mixtral_quant.py:137-157) spends ten small framework kernels
// on the routing weights (softmax in fp32, top-k, renormalise, scatter) and, per expert, a multiply and an add
// on the output; around two batched W4A16 launches of ~20 us each that is most of a decode step's expert block.
// Both kernels are HBM-trivial (a few KiB in, a few KiB out): what they save is launches.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/w4a16.h"

namespace {

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kMaxExperts = 64;  // one pass of a 64-bit selection mask
constexpr int kRouteThreads = 256;

// One thread per token (a workgroup walks all tokens, so that the per-expert "somebody chose me" flags need no
// second launch): softmax over all experts in fp32 as torch.softmax(dtype=float) takes it (max, exp, sum, divide),
// the top_k largest (ties: the lower index first), renormalised by their sum, written densely for the local experts.
__global__ void __launch_bounds__(kRouteThreads) moe_route_kernel(const f16* __restrict__ logits, float* __restrict__ weights,
                                                                  int* __restrict__ mask, int tokens, int n_experts, int top_k,
                                                                  int first_local, int n_local) {
  __shared__ int chosen[kMaxExperts];
  if (threadIdx.x < kMaxExperts) chosen[threadIdx.x] = 0;
  __syncthreads();
  for (int t = threadIdx.x; t < tokens; t += kRouteThreads) {
    const f16* row = logits + (size_t)t * n_experts;
    float p[kMaxExperts];
    float mx = -INFINITY;
    for (int e = 0; e < n_experts; ++e) {
      p[e] = (float)row[e];
      mx = fmaxf(mx, p[e]);
    }
    float sum = 0.f;
    for (int e = 0; e < n_experts; ++e) {
      p[e] = expf(p[e] - mx);
      sum += p[e];
    }
    for (int e = 0; e < n_experts; ++e) p[e] /= sum;
    unsigned long long taken = 0;
    float top_sum = 0.f;
    for (int k = 0; k < top_k; ++k) {
      int best = -1;
      for (int e = 0; e < n_experts; ++e)
        if (!((taken >> e) & 1ull) && (best < 0 || p[e] > p[best])) best = e;
      taken |= 1ull << best;
      top_sum += p[best];
    }
    for (int j = 0; j < n_local; ++j) {
      const int e = first_local + j;
      const bool on = (taken >> e) & 1ull;
      const float w = on ? p[e] / top_sum : 0.f;
      weights[(size_t)t * n_local + j] = w;
      if (w != 0.f) chosen[j] = 1;  // (benign race: every writer stores 1)
    }
  }
  __syncthreads();
  if (mask != nullptr && threadIdx.x < n_local) mask[threadIdx.x] = chosen[threadIdx.x];
}

// out[t, h] = fp16(sum over local experts e of float(fp16(float(y[e, t, h]) * w[t, e]))): each expert's output is
// scaled and rounded to fp16 as `current_hidden_states.mul_(expert_weights)` rounds it (mixtral_quant.py:152-153),
// the sum over experts runs in fp32 in expert order and is rounded once.  Experts with weight 0 for a token are
// not read at all: what a masked-out problem left in y never reaches the sum.
// One thread per 8 columns of one token.
__global__ void __launch_bounds__(256) moe_combine_kernel(const f16* __restrict__ y, const float* __restrict__ weights,
                                                          f16* __restrict__ out, int n_local, int tokens, int hidden) {
  const int chunks = hidden / 8;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)tokens * chunks) return;
  const int t = (int)(idx / chunks), c = (int)(idx - (long long)t * chunks);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int e = 0; e < n_local; ++e) {
    const float w = weights[(size_t)t * n_local + e];
    if (w == 0.f) continue;
    const f16x8 v = *reinterpret_cast<const f16x8*>(y + ((size_t)e * tokens + t) * hidden + (size_t)c * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += (float)(f16)((float)v[i] * w);
  }
  f16x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (f16)acc[i];
  *reinterpret_cast<f16x8*>(out + (size_t)t * hidden + (size_t)c * 8) = r;
}

}  // namespace

extern "C" __attribute__((visibility("hidden"))) void w4a16_note_hip_error(int e);  // w4a16.hip

namespace {
int after_launch() {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return W4A16_OK;
  w4a16_note_hip_error((int)e);
  return W4A16_E_LAUNCH;
}
}  // namespace

extern "C" int w4a16_moe_route_f16(const void* router_logits, void* weights, void* mask, int tokens, int n_experts,
                                   int top_k, int first_local, int n_local, void* stream) {
  if (!router_logits || !weights) return W4A16_E_NULL;
  if (tokens <= 0 || n_experts <= 0 || n_experts > kMaxExperts || top_k <= 0 || top_k > n_experts) return W4A16_E_SHAPE;
  if (first_local < 0 || n_local <= 0 || first_local + n_local > n_experts) return W4A16_E_SHAPE;
  hipLaunchKernelGGL(moe_route_kernel, dim3(1), dim3(kRouteThreads), 0, static_cast<hipStream_t>(stream),
                     static_cast<const f16*>(router_logits), static_cast<float*>(weights), static_cast<int*>(mask), tokens,
                     n_experts, top_k, first_local, n_local);
  return after_launch();
}

extern "C" int w4a16_moe_combine_f16(const void* y, const void* weights, void* out, int n_local, int tokens, int hidden,
                                     void* stream) {
  if (!y || !weights || !out) return W4A16_E_NULL;
  if (n_local <= 0 || tokens <= 0 || hidden <= 0 || hidden % 8) return W4A16_E_SHAPE;
  if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out)) & 15u) return W4A16_E_ALIGN;
  const long long items = (long long)tokens * (hidden / 8);
  if ((items + 255) / 256 >= (1LL << 31)) return W4A16_E_TOO_LARGE;
  hipLaunchKernelGGL(moe_combine_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const f16*>(y), static_cast<const float*>(weights),
                     static_cast<f16*>(out), n_local, tokens, hidden);
  return after_launch();
}
