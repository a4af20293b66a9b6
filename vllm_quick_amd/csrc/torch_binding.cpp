// Native PyTorch binding of the AWQ operators: the compiled counterpart of the reference's
//   ops.def("awq_gemm", &awq_gemm, ...);  ops.def("awq_dequantize", &awq_dequantize, ...);
// (csrc/pybind.cpp:52-55, declarations csrc/ops.h:69-82, bodies csrc/quantization/awq/gemm_kernels.cu:
// 332-381 and 389-446), built as the extension module vllm_quick_amd._C_native with a submodule `ops`.
//
// Same signatures, same result shape / dtype / device, same exception types (std::invalid_argument
// for the shape checks -> ValueError; c10::Error for dtype / device mismatches -> RuntimeError), the
// device guard and torch's current stream as the reference takes them (gemm_kernels.cu:398, :421).
// All arithmetic is done by libw4a16.so through its C ABI (include/w4a16.h): this file holds no kernel.
//
// awq_gemm is handed the STOCK tensors on every call.  The first call for a layer re-orders them once
// into the library's private layout (w4a16_repack) and keeps the result in a process-wide table keyed
// by the three storages; later calls go straight to the matrix-instruction kernels.  An entry is reused
// only for the very same tensors at the same version counters, so in-place writes PyTorch counts
// (copy_, add_, ... on the tensor itself), freed and re-used addresses and fresh views all miss and
// re-pack.  Two kinds of write are invisible to a version counter:
//  * writes through `param.data` (the reference's weight loaders: `param.data.copy_(..)` leaves
//    `param._version` alone): after such a load that follows a first call, the caller must call
//    clear_repack_cache() (the layers of this package do the equivalent in their own loaders);
//  * tensors created under torch.inference_mode() keep no counter at all: their packed copies are NOT kept
//    (every call re-packs, warned about once), so an in-place reload is always seen.
// The re-layout runs on the stream of the call that needs it, and that stream is synchronised once before
// the entry becomes visible: calls on other streams never see a half-written copy.
// Nothing is re-ordered or allocated for the table while the current stream is capturing a graph: such
// a call is served by the shape-agnostic kernel on the stock layout (three to five times slower), counted
// in stock_kernel_calls() and warned about once - run one eager warm-up call per layer before capturing,
// as vLLM does (vllm/worker/model_runner.py:765-775).
#include <ATen/hip/HIPContext.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/extension.h>

#include <atomic>
#include <mutex>
#include <stdexcept>
#include <tuple>
#include <unordered_map>

#include "../../include/w4a16.h"

namespace {

using c10::hip::OptionalHIPGuardMasqueradingAsCUDA;
using c10::hip::getCurrentHIPStreamMasqueradingAsCUDA;

constexpr int kPackFactor = 8;

struct Geometry {
  int K, N, G;
};

void raise_for(int rc, const char* what) {
  if (rc == W4A16_OK) return;
  const std::string msg = std::string(what) + ": " + w4a16_strerror(rc);
  switch (rc) {
    case W4A16_E_OC_PACK:
    case W4A16_E_GROUP32:
    case W4A16_E_K_GROUP:
    case W4A16_E_SHAPE:
      throw std::invalid_argument(msg);  // as gemm_kernels.cu:412-419
    default:
      TORCH_CHECK(false, msg, rc == W4A16_E_LAUNCH ? " (hipError " + std::to_string(w4a16_last_hip_error()) + ")" : "");
  }
}

// The checks of gemm_kernels.cu:405-419 that do not depend on its launch shape (OC % 64 is not required
// here: 1376-wide tensor-parallel shards are served), plus the consistency checks raw pointers need.
Geometry geometry(const torch::Tensor& kernel, const torch::Tensor& scaling_factors, const torch::Tensor& zeros) {
  TORCH_CHECK(kernel.scalar_type() == at::kInt, "expected scalar type Int but found ", kernel.scalar_type(), " for kernel");
  TORCH_CHECK(scaling_factors.scalar_type() == at::kHalf, "expected scalar type Half but found ",
              scaling_factors.scalar_type(), " for scaling_factors");
  TORCH_CHECK(zeros.scalar_type() == at::kInt, "expected scalar type Int but found ", zeros.scalar_type(), " for zeros");
  TORCH_CHECK(kernel.is_cuda() && scaling_factors.is_cuda() && zeros.is_cuda(),
              "kernel, scaling_factors and zeros must be GPU tensors: the W4A16 path has no CPU implementation");
  TORCH_CHECK(kernel.device() == scaling_factors.device() && kernel.device() == zeros.device(),
              "kernel, scaling_factors and zeros must be on the same device");
  if (kernel.dim() != 2 || scaling_factors.dim() != 2 || zeros.dim() != 2)
    throw std::invalid_argument("kernel, scaling_factors and zeros must be 2-D");
  const int64_t K = kernel.size(0), NC = kernel.size(1), groups = scaling_factors.size(0);
  if (groups == 0 || K % groups != 0) throw std::invalid_argument("IC is not multiple of Group size");
  if (scaling_factors.size(1) != NC * kPackFactor) throw std::invalid_argument("scaling_factors must be [IC / G, OC]");
  if (zeros.size(0) != groups || zeros.size(1) != NC) throw std::invalid_argument("zeros must be [IC / G, OC / 8]");
  const int64_t G = K / groups;
  if (G % 32 != 0) throw std::invalid_argument("Group size should be a multiple of 32");
  return Geometry{(int)K, (int)(NC * kPackFactor), (int)G};
}

uint32_t version_of(const torch::Tensor& t) {
  // (inference-mode tensors carry no version counter: packed_for never keeps an entry for them)
  return t.is_inference() ? 0u : t._version();
}

struct Packed {
  torch::Tensor w, sz;  // uint8 buffers in the private layout
  c10::weak_intrusive_ptr<c10::TensorImpl> refs[3];
  uint32_t versions[3];
  Packed(torch::Tensor w_, torch::Tensor sz_, const torch::Tensor* src[3])
      : w(std::move(w_)), sz(std::move(sz_)),
        refs{c10::weak_intrusive_ptr<c10::TensorImpl>(src[0]->getIntrusivePtr()),
             c10::weak_intrusive_ptr<c10::TensorImpl>(src[1]->getIntrusivePtr()),
             c10::weak_intrusive_ptr<c10::TensorImpl>(src[2]->getIntrusivePtr())},
        versions{version_of(*src[0]), version_of(*src[1]), version_of(*src[2])} {}
  bool matches(const torch::Tensor* src[3]) const {
    for (int i = 0; i < 3; ++i) {
      auto alive = refs[i].lock();
      if (!alive || alive.get() != src[i]->unsafeGetTensorImpl() || versions[i] != version_of(*src[i])) return false;
    }
    return true;
  }
  bool dead() const { return refs[0].expired() || refs[1].expired() || refs[2].expired(); }
};

using Key = std::tuple<const void*, const void*, const void*, int64_t, int64_t, int64_t, int>;
struct KeyHash {
  size_t operator()(const Key& k) const {
    size_t h = std::hash<const void*>()(std::get<0>(k));
    h = h * 1000003u ^ std::hash<const void*>()(std::get<1>(k));
    h = h * 1000003u ^ std::hash<const void*>()(std::get<2>(k));
    h = h * 1000003u ^ std::hash<int64_t>()(std::get<3>(k) * 131 + std::get<4>(k) * 17 + std::get<5>(k));
    return h * 1000003u ^ (size_t)std::get<6>(k);
  }
};

std::mutex g_mutex;
std::unordered_map<Key, std::shared_ptr<Packed>, KeyHash> g_table;
std::atomic<long long> g_stock_calls{0}, g_repacks{0};
constexpr size_t kTableLimit = 4096;

bool capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}

// The layer's weights in the private layout, or nullptr when they are not there and must not be made now.
std::shared_ptr<Packed> packed_for(const torch::Tensor& kernel, const torch::Tensor& scaling_factors,
                                   const torch::Tensor& zeros, const Geometry& g, hipStream_t stream) {
  const torch::Tensor* src[3] = {&kernel, &scaling_factors, &zeros};
  // Without version counters an in-place reload cannot be told from no change: nothing is kept for such tensors.
  const bool trackable = !(kernel.is_inference() || scaling_factors.is_inference() || zeros.is_inference());
  if (!trackable) {
    if (capturing(stream)) return nullptr;
    TORCH_WARN_ONCE(
        "awq_gemm: the weights are inference-mode tensors, which keep no version counter: they are re-ordered "
        "into the private layout on every call; create weights outside torch.inference_mode() to have that done once");
    const size_t wbytes = w4a16_packed_weight_bytes(g.K, g.N), sbytes = w4a16_packed_sz_bytes(g.K, g.N, g.G);
    if (wbytes == 0 || sbytes == 0) throw std::invalid_argument("cannot repack these shapes");
    const auto opts = torch::TensorOptions().dtype(torch::kUInt8).device(kernel.device());
    auto pw = torch::empty({(int64_t)wbytes}, opts), ps = torch::empty({(int64_t)sbytes}, opts);
    raise_for(w4a16_repack(kernel.data_ptr(), scaling_factors.data_ptr(), zeros.data_ptr(), pw.data_ptr(), ps.data_ptr(),
                           g.K, g.N, g.G, stream),
              "awq_gemm (repack)");
    ++g_repacks;
    return std::make_shared<Packed>(std::move(pw), std::move(ps), src);  // used on this stream, then dropped
  }
  const Key key{kernel.data_ptr(), scaling_factors.data_ptr(), zeros.data_ptr(), kernel.size(0), kernel.size(1),
                scaling_factors.size(0), (int)kernel.get_device()};
  std::lock_guard<std::mutex> lock(g_mutex);
  auto it = g_table.find(key);
  if (it != g_table.end() && it->second->matches(src)) return it->second;
  if (capturing(stream)) return nullptr;
  if (g_table.size() >= kTableLimit) {
    for (auto e = g_table.begin(); e != g_table.end();) e = e->second->dead() ? g_table.erase(e) : std::next(e);
    if (g_table.size() >= kTableLimit) g_table.clear();
  }
  const size_t wbytes = w4a16_packed_weight_bytes(g.K, g.N), sbytes = w4a16_packed_sz_bytes(g.K, g.N, g.G);
  if (wbytes == 0 || sbytes == 0) throw std::invalid_argument("cannot repack these shapes");
  const auto opts = torch::TensorOptions().dtype(torch::kUInt8).device(kernel.device());
  auto pw = torch::empty({(int64_t)wbytes}, opts), ps = torch::empty({(int64_t)sbytes}, opts);
  raise_for(w4a16_repack(kernel.data_ptr(), scaling_factors.data_ptr(), zeros.data_ptr(), pw.data_ptr(), ps.data_ptr(),
                         g.K, g.N, g.G, stream),
            "awq_gemm (repack)");
  ++g_repacks;
  // once per layer: the copy is complete before any other stream can find it in the table
  TORCH_CHECK(hipStreamSynchronize(stream) == hipSuccess, "awq_gemm (repack): hipStreamSynchronize failed");
  auto entry = std::make_shared<Packed>(std::move(pw), std::move(ps), src);
  g_table[key] = entry;
  return entry;
}

// csrc/ops.h:69-74, gemm_kernels.cu:389-446
torch::Tensor awq_gemm(torch::Tensor _in_feats, torch::Tensor _kernel, torch::Tensor _scaling_factors,
                       torch::Tensor _zeros, int split_k_iters) {
  const Geometry g = geometry(_kernel, _scaling_factors, _zeros);
  // bf16 activations are an extension (the reference takes fp16 only): they run on the packed copy alone
  const bool bf16 = _in_feats.scalar_type() == at::kBFloat16;
  TORCH_CHECK(bf16 || _in_feats.scalar_type() == at::kHalf, "expected scalar type Half but found ",
              _in_feats.scalar_type(), " for in_feats");
  TORCH_CHECK(_in_feats.is_cuda() && _in_feats.device() == _kernel.device(),
              "in_feats must be a GPU tensor on the device of the weights");
  if (_in_feats.dim() != 2 || _in_feats.size(1) != g.K) throw std::invalid_argument("in_feats must be [M, IC]");
  const OptionalHIPGuardMasqueradingAsCUDA device_guard(device_of(_in_feats));
  const hipStream_t stream = getCurrentHIPStreamMasqueradingAsCUDA().stream();
  const torch::Tensor x = _in_feats.contiguous();  // the reference reads raw data_ptr and assumes this (:405)
  const int M = (int)x.size(0);
  torch::Tensor out = torch::empty({M, g.N}, x.options());
  if (M == 0) return out;
  std::shared_ptr<Packed> packed;
  if (_kernel.is_contiguous() && _scaling_factors.is_contiguous() && _zeros.is_contiguous())
    packed = packed_for(_kernel, _scaling_factors, _zeros, g, stream);
  if (packed && bf16) {
    raise_for(w4a16_gemm_packed_bf16(x.data_ptr(), packed->w.data_ptr(), packed->sz.data_ptr(), nullptr,
                                     out.data_ptr(), M, g.K, g.N, g.G, W4A16_EPI_NONE, 0, 0, 0, stream),
              "awq_gemm (bf16)");
    return out;
  }
  if (packed) {
    raise_for(w4a16_gemm_packed_f16(x.data_ptr(), packed->w.data_ptr(), packed->sz.data_ptr(), nullptr, out.data_ptr(),
                                    M, g.K, g.N, g.G, stream),
              "awq_gemm");
    return out;
  }
  TORCH_CHECK(!bf16, "awq_gemm: bf16 activations need the packed copy of the weights, which cannot be made now "
                     "(non-contiguous weights, or a first call on these weights under graph capture)");
  ++g_stock_calls;
  TORCH_WARN_ONCE(
      "awq_gemm: serving a call with the shape-agnostic kernel on the stock layout (non-contiguous weights, or a "
      "first call under graph capture); run one eager call per layer first to get the matrix-instruction kernels");
  const torch::Tensor k = _kernel.contiguous(), s = _scaling_factors.contiguous(), z = _zeros.contiguous();
  raise_for(w4a16_gemm_f16(x.data_ptr(), k.data_ptr(), s.data_ptr(), z.data_ptr(), out.data_ptr(), M, g.K, g.N, g.G,
                           split_k_iters, stream),
            "awq_gemm");
  return out;
}

// csrc/ops.h:76-82, gemm_kernels.cu:332-381 (split_k_iters, thx, thy only shaped the reference's launch)
torch::Tensor awq_dequantize(torch::Tensor _kernel, torch::Tensor _scaling_factors, torch::Tensor _zeros,
                             int split_k_iters, int thx, int thy) {
  (void)split_k_iters, (void)thx, (void)thy;
  const Geometry g = geometry(_kernel, _scaling_factors, _zeros);
  const OptionalHIPGuardMasqueradingAsCUDA device_guard(device_of(_kernel));
  const hipStream_t stream = getCurrentHIPStreamMasqueradingAsCUDA().stream();
  const torch::Tensor k = _kernel.contiguous(), s = _scaling_factors.contiguous(), z = _zeros.contiguous();
  torch::Tensor out = torch::empty({g.K, g.N}, s.options());
  raise_for(w4a16_dequantize_f16(k.data_ptr(), s.data_ptr(), z.data_ptr(), out.data_ptr(), g.K, g.N, g.G, stream),
            "awq_dequantize");
  return out;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "AWQ W4A16 operators on libw4a16.so (MI355X)";
  pybind11::module ops = m.def_submodule("ops", "vLLM custom operators");
  ops.def("awq_gemm", &awq_gemm, "Quantized GEMM for AWQ");
  ops.def("awq_dequantize", &awq_dequantize, "Dequantization for AWQ");
  m.def("stock_kernel_calls", [] { return g_stock_calls.load(); },
        "awq_gemm calls served by the shape-agnostic stock-layout kernel since the module was loaded");
  m.def("repack_count", [] { return g_repacks.load(); }, "layers re-ordered into the private layout so far");
  m.def("clear_repack_cache", [] {
    std::lock_guard<std::mutex> lock(g_mutex);
    g_table.clear();
  });
  m.def("abi_version", [] { return w4a16_abi_version(); });
}
