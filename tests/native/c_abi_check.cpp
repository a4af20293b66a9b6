// A host program with no PyTorch in it: drives libw4a16.so through include/w4a16.h with plain HIP
// allocations, the way a C/C++ caller of the reference's operators would, and checks the results
// against the C restatement of the reference arithmetic (oracle/awq_oracle.c - the checker; this
// program is test infrastructure, built by __graft_entry__.build() and run by tests/test_native_gpu.py).
//
//   c_abi_check            -> prints one line per check and exits 0 if all pass
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "w4a16.h"

extern "C" {
void awq_c_dequantize(const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros, uint16_t* out, int K,
                      int N, int G);
int awq_c_gemm(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
               uint16_t* out, int M, int K, int N, int G, int split_k, int faithful);
uint16_t awq_c_f2h(float f);
float awq_c_h2f(uint16_t h);
}

#define HIP_OK(call)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      std::fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));           \
      std::exit(2);                                                                    \
    }                                                                                  \
  } while (0)

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() {  // xorshift64*
  rng_state ^= rng_state >> 12;
  rng_state ^= rng_state << 25;
  rng_state ^= rng_state >> 27;
  return (uint32_t)((rng_state * 0x2545f4914f6cdd1dull) >> 32);
}
static float uniform() { return (rnd() >> 8) * (1.0f / 16777216.0f); }
static float normal() {
  const float u1 = uniform() + 1e-7f, u2 = uniform();
  return std::sqrt(-2.0f * std::log(u1)) * std::cos(6.2831853f * u2);
}

template <typename T>
static T* to_device(const std::vector<T>& h) {
  T* d = nullptr;
  HIP_OK(hipMalloc(&d, h.size() * sizeof(T)));
  HIP_OK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

static int check_case(int M, int K, int N, int G, bool with_bias) {
  std::vector<int32_t> qw((size_t)K * N / 8), qz((size_t)(K / G) * N / 8);
  std::vector<uint16_t> sc((size_t)(K / G) * N), x((size_t)M * K), bias(N);
  for (auto& v : qw) v = (int32_t)rnd();
  for (auto& v : qz) v = (int32_t)rnd();
  for (auto& v : sc) v = awq_c_f2h(1e-3f + uniform() * 1.9e-2f);
  for (auto& v : x) v = awq_c_f2h(normal() * 0.25f);
  for (auto& v : bias) v = awq_c_f2h(normal() * 0.1f);

  int32_t *d_qw = to_device(qw), *d_qz = to_device(qz);
  uint16_t *d_sc = to_device(sc), *d_x = to_device(x), *d_bias = to_device(bias);
  void *d_pw = nullptr, *d_ps = nullptr;
  uint16_t *d_out = nullptr, *d_deq = nullptr;
  HIP_OK(hipMalloc(&d_pw, w4a16_packed_weight_bytes(K, N)));
  HIP_OK(hipMalloc(&d_ps, w4a16_packed_sz_bytes(K, N, G)));
  HIP_OK(hipMalloc(&d_out, (size_t)M * N * 2));
  HIP_OK(hipMalloc(&d_deq, (size_t)K * N * 2));
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));

  int fails = 0;
  auto expect_ok = [&](int rc, const char* what) {
    if (rc != W4A16_OK) {
      std::printf("FAIL %s: %s (hip error %d)\n", what, w4a16_strerror(rc), w4a16_last_hip_error());
      ++fails;
    }
  };
  // oracle results
  std::vector<uint16_t> want_deq((size_t)K * N), want((size_t)M * N), got((size_t)M * N), got_deq((size_t)K * N);
  awq_c_dequantize(qw.data(), sc.data(), qz.data(), want_deq.data(), K, N, G);
  if (awq_c_gemm(x.data(), qw.data(), sc.data(), qz.data(), want.data(), M, K, N, G, 8, 1) != 0) return 1;

  auto compare = [&](const char* what, bool add_bias) {
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(got.data(), d_out, got.size() * 2, hipMemcpyDeviceToHost));
    size_t bad = 0;
    double worst = 0;
    for (size_t i = 0; i < got.size(); ++i) {
      double w = awq_c_h2f(want[i]);
      if (add_bias) w = awq_c_h2f(awq_c_f2h((float)w + awq_c_h2f(bias[i % N])));
      const double d = std::fabs((double)awq_c_h2f(got[i]) - w);
      worst = d > worst ? d : worst;
      if (d > 1e-3 * (add_bias ? 2 : 1) + 1e-2 * std::fabs(w)) ++bad;  // rtol 1e-2 / atol 1e-3 (2e-3 with the fp16 bias add)
    }
    std::printf("%s %s M=%d K=%d N=%d G=%d: %zu of %zu outside tolerance, max |d| %.3e\n", bad ? "FAIL" : "ok  ",
                what, M, K, N, G, bad, got.size(), worst);
    fails += bad != 0;
  };

  expect_ok(w4a16_gemm_f16(d_x, d_qw, d_sc, d_qz, d_out, M, K, N, G, 8, stream), "w4a16_gemm_f16");
  compare("gemm (stock layout)   ", false);
  expect_ok(w4a16_repack(d_qw, d_sc, d_qz, d_pw, d_ps, K, N, G, stream), "w4a16_repack");
  HIP_OK(hipMemsetAsync(d_out, 0xff, (size_t)M * N * 2, stream));
  expect_ok(w4a16_gemm_packed_f16(d_x, d_pw, d_ps, with_bias ? d_bias : nullptr, d_out, M, K, N, G, stream),
            "w4a16_gemm_packed_f16");
  compare(with_bias ? "gemm (packed) + bias  " : "gemm (packed)         ", with_bias);

  {
    // add-to-out epilogue: residual + (the result just checked), bit for bit the fp16 sum
    std::vector<uint16_t> plain = got, res((size_t)M * N), sum((size_t)M * N);
    for (auto& v : res) v = awq_c_f2h(normal());
    HIP_OK(hipMemcpyAsync(d_out, res.data(), res.size() * 2, hipMemcpyHostToDevice, stream));
    expect_ok(w4a16_gemm_packed_f16_ex(d_x, 0, d_pw, d_ps, with_bias ? d_bias : nullptr, d_out, (long long)M * N, M, K,
                                       N, G, 1, W4A16_EPI_ADD_TO_OUT, 0, 0, 0, 0, stream),
              "w4a16_gemm_packed_f16_ex(ADD_TO_OUT)");
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(sum.data(), d_out, sum.size() * 2, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < sum.size(); ++i) diff += sum[i] != awq_c_f2h(awq_c_h2f(res[i]) + awq_c_h2f(plain[i]));
    std::printf("%s gemm (packed) += out   M=%d K=%d N=%d G=%d: %zu elements differ from the fp16 sum\n",
                diff ? "FAIL" : "ok  ", M, K, N, G, diff);
    fails += diff != 0;
  }

  if (M <= 4 && K % 128 == 0 && K >= 256) {
    // The decode kernel with K split over two workgroups per output slot: on a workspace from hipMalloc (zeroed
    // once), then on the library's own device array, which needs no allocation at all; same launch, same bits.
    const size_t ws_bytes = w4a16_decode_workspace_bytes(K, N, 1, W4A16_EPI_NONE, 1, 8, 2);
    void* d_ws = nullptr;
    HIP_OK(hipMalloc(&d_ws, ws_bytes));
    HIP_OK(hipMemsetAsync(d_ws, 0, ws_bytes, stream));
    std::vector<uint16_t> first((size_t)M * N);
    for (int round = 0; round < 2; ++round) {
      HIP_OK(hipMemsetAsync(d_out, 0xff, (size_t)M * N * 2, stream));
      expect_ok(round == 0 ? w4a16_gemm_packed_f16_decode_cfg(d_x, 0, d_pw, d_ps, with_bias ? d_bias : nullptr, d_out, 0,
                                                              M, K, N, G, 1, W4A16_EPI_NONE, 1, 8, 0, 2, d_ws, ws_bytes,
                                                              stream)
                           : w4a16_gemm_packed_f16_decode_static(d_x, 0, d_pw, d_ps, with_bias ? d_bias : nullptr, d_out,
                                                                 0, M, K, N, G, 1, W4A16_EPI_NONE, 1, 8, 0, 2, stream),
                round == 0 ? "w4a16_gemm_packed_f16_decode_cfg" : "w4a16_gemm_packed_f16_decode_static");
      compare(round == 0 ? "decode, K split, workspace  " : "decode, K split, static     ", with_bias);
      if (round == 0) first = got;
    }
    size_t diff = 0;
    for (size_t i = 0; i < got.size(); ++i) diff += got[i] != first[i];
    std::printf("%s decode static vs workspace: %zu elements differ\n", diff ? "FAIL" : "ok  ", diff);
    fails += diff != 0;
    if (w4a16_gemm_packed_f16_decode_cfg(d_x, 0, d_pw, d_ps, nullptr, d_out, 0, M, K, N, G, 1, W4A16_EPI_NONE, 1, 8, 0, 2,
                                         d_ws, ws_bytes - 16, stream) != W4A16_E_WORKSPACE)
      ++fails;
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipFree(d_ws));
  }

  for (int packed = 0; packed < 2; ++packed) {
    expect_ok(packed ? w4a16_dequantize_packed_f16(d_pw, d_ps, d_deq, K, N, G, stream)
                     : w4a16_dequantize_f16(d_qw, d_sc, d_qz, d_deq, K, N, G, stream),
              "dequantize");
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(got_deq.data(), d_deq, got_deq.size() * 2, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < got_deq.size(); ++i) diff += got_deq[i] != want_deq[i];
    std::printf("%s dequantize (%s) K=%d N=%d G=%d: %zu words differ (must be bit-exact)\n", diff ? "FAIL" : "ok  ",
                packed ? "packed" : "stock ", K, N, G, diff);
    fails += diff != 0;
  }
  // argument errors come back as codes, nothing is launched
  if (w4a16_gemm_f16(d_x, d_qw, d_sc, d_qz, d_out, M, K, N + 4, G, 8, stream) != W4A16_E_OC_PACK) ++fails;
  if (w4a16_gemm_packed_f16(nullptr, d_pw, d_ps, nullptr, d_out, M, K, N, G, stream) != W4A16_E_NULL) ++fails;

  HIP_OK(hipStreamDestroy(stream));
  for (void* p : {(void*)d_qw, (void*)d_qz, (void*)d_sc, (void*)d_x, (void*)d_bias, d_pw, d_ps, (void*)d_out, (void*)d_deq})
    HIP_OK(hipFree(p));
  return fails;
}

// Expert routing and the weighted sum over experts against a host restatement of
// vllm/model_executor/models/mixtral_quant.py:137-157 (softmax in fp32, top-2, renormalise; each expert's
// output scaled and rounded to fp16, the sum taken in fp32 and rounded once).
static int check_routing(int tokens, int n_experts, int first, int n_local, int hidden) {
  const int top_k = 2;
  std::vector<uint16_t> logits((size_t)tokens * n_experts), y((size_t)n_local * tokens * hidden);
  for (auto& v : logits) v = awq_c_f2h(2.0f * normal());
  for (auto& v : y) v = awq_c_f2h(3.0f * normal());
  std::vector<float> want_w((size_t)tokens * n_local, 0.f);
  std::vector<int> want_mask(n_local, 0);
  for (int t = 0; t < tokens; ++t) {
    std::vector<float> p(n_experts);
    float mx = -INFINITY, sum = 0.f;
    for (int e = 0; e < n_experts; ++e) mx = std::fmax(mx, p[e] = awq_c_h2f(logits[(size_t)t * n_experts + e]));
    for (int e = 0; e < n_experts; ++e) sum += p[e] = std::exp(p[e] - mx);
    for (int e = 0; e < n_experts; ++e) p[e] /= sum;
    int a = 0;
    for (int e = 1; e < n_experts; ++e) if (p[e] > p[a]) a = e;
    int b = a == 0 ? 1 : 0;
    for (int e = 0; e < n_experts; ++e) if (e != a && p[e] > p[b]) b = e;
    for (int e : {a, b})
      if (e >= first && e < first + n_local) {
        want_w[(size_t)t * n_local + e - first] = p[e] / (p[a] + p[b]);
        want_mask[e - first] = 1;
      }
  }
  uint16_t* d_logits = to_device(logits);
  uint16_t* d_y = to_device(y);
  float* d_w = nullptr;
  int* d_mask = nullptr;
  uint16_t* d_out = nullptr;
  HIP_OK(hipMalloc(&d_w, want_w.size() * 4));
  HIP_OK(hipMalloc(&d_mask, n_local * 4));
  HIP_OK(hipMalloc(&d_out, (size_t)tokens * hidden * 2));
  int fails = 0;
  int rc = w4a16_moe_route_f16(d_logits, d_w, d_mask, tokens, n_experts, top_k, first, n_local, nullptr);
  if (rc == W4A16_OK) rc = w4a16_moe_combine_f16(d_y, d_w, d_out, n_local, tokens, hidden, nullptr);
  if (rc != W4A16_OK) {
    std::printf("FAIL routing: %s (hip error %d)\n", w4a16_strerror(rc), w4a16_last_hip_error());
    return 1;
  }
  HIP_OK(hipDeviceSynchronize());
  std::vector<float> w(want_w.size());
  std::vector<int> mask(n_local);
  std::vector<uint16_t> out((size_t)tokens * hidden);
  HIP_OK(hipMemcpy(w.data(), d_w, w.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(mask.data(), d_mask, n_local * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(out.data(), d_out, out.size() * 2, hipMemcpyDeviceToHost));
  size_t bad_w = 0, bad_mask = 0, bad_out = 0;
  for (size_t i = 0; i < w.size(); ++i)
    bad_w += ((w[i] == 0.f) != (want_w[i] == 0.f)) || std::fabs(w[i] - want_w[i]) > 2e-6f * want_w[i] + 1e-9f;
  for (int e = 0; e < n_local; ++e) bad_mask += mask[e] != want_mask[e];
  // the sum over experts from the DEVICE's weights: two terms, so bit-exact whatever the order
  for (int t = 0; t < tokens; ++t)
    for (int h = 0; h < hidden; ++h) {
      float acc = 0.f;
      for (int e = 0; e < n_local; ++e) {
        const float we = w[(size_t)t * n_local + e];
        if (we != 0.f) acc += awq_c_h2f(awq_c_f2h(awq_c_h2f(y[((size_t)e * tokens + t) * hidden + h]) * we));
      }
      bad_out += out[(size_t)t * hidden + h] != awq_c_f2h(acc);
    }
  fails = (bad_w || bad_mask || bad_out) ? 1 : 0;
  std::printf("%s routing + weighted sum  tokens=%d experts=%d local=[%d,%d) hidden=%d: %zu weights, %zu mask entries, "
              "%zu outputs differ\n", fails ? "FAIL" : "ok  ", tokens, n_experts, first, first + n_local, hidden, bad_w,
              bad_mask, bad_out);
  for (void* p : {(void*)d_logits, (void*)d_y, (void*)d_w, (void*)d_mask, (void*)d_out}) HIP_OK(hipFree(p));
  return fails;
}

int main() {
  if (w4a16_abi_version() != W4A16_ABI_VERSION) {
    std::printf("FAIL ABI version %d, header says %d\n", w4a16_abi_version(), W4A16_ABI_VERSION);
    return 1;
  }
  int fails = 0;
  fails += check_case(1, 512, 256, 128, false);
  fails += check_case(3, 1536, 272, 128, true);
  fails += check_case(7, 384, 200, 64, true);
  fails += check_case(40, 640, 416, 32, false);
  fails += check_case(130, 1024, 1376, 128, true);
  fails += check_routing(1, 8, 0, 8, 256);
  fails += check_routing(37, 8, 4, 4, 512);
  std::printf(fails ? "FAILED (%d)\n" : "ALL OK\n", fails);
  return fails ? 1 : 0;
}
