// libp2p_ar.so: push-based all-reduce(sum) between the GPUs of one node (gfx950, xGMI full mesh).
//
// Memory: every rank allocates one mailbox with hipExtMallocWithFlags(hipDeviceMallocUncached),
// i.e. fine-grained memory that no L2 keeps lines of, so that what a peer GPU writes into it is
// what the owning GPU reads next, inside one running kernel.  Layout of a mailbox:
//
//   flags  [2 halves][2 stages][8 source ranks][kMaxBlocks]  u32      (kFlagBytes)
//   data   [2 halves][region_bytes]                                    16-byte units
//          one-stage: slot of source rank s at unit s * units
//          two-stage: stage-1 slots at unit s * slot_units, stage-2 slots half_bytes further on
//
// Protocol of call number e (e = 1, 2, ..; the counter lives in device memory, the kernel's last
// workgroup advances it, so graph replays and eager calls share one sequence): half = e & 1.
// Workgroup b of rank r writes its share of r's input into slot r of every peer's half, makes
// the writes visible (system-scope release fence), stores e into flag[half][stage][r][b] of every
// peer, then waits until flag[half][stage][s][b] of its own mailbox reaches e for every peer s
// and sums slot by slot from its own mailbox.  Only same-numbered workgroups of different ranks
// depend on each other, and all ranks launch the same grid for the same call, so a rank needs no
// grid-wide barrier.  No end-of-call barrier either: a rank can run at most one call ahead of a
// peer (finishing call e needs every peer's flags of call e, which a peer raises after it has
// finished call e-1), and call e+1 uses the other half.
//
// Every wait is bounded (wall clock and an iteration cap): a rank whose peer never arrives sets
// the sticky status word, in device memory and in a pinned host word the host reads without
// synchronising, and carries on, so a kernel of this library always terminates and the host learns
// of the invalid result at its next call.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdint>
#include <cstring>
#include <new>

#include "../../include/p2p_allreduce.h"

namespace {

constexpr int kMaxRanks = 8;
constexpr int kMaxBlocks = 64;
constexpr int kThreads = 512;
constexpr int kOneStageBlocks = 32;
constexpr int64_t kFlagBytes = 2 * 2 * kMaxRanks * kMaxBlocks * 4;
constexpr unsigned long long kSpinCap = 1ull << 34;  // backstop only: hours; the wall clock decides

thread_local int g_last_hip_error = 0;

struct Ctx {
  int rank = 0, world = 0, device = 0;
  int64_t max_bytes = 0, half_bytes = 0, region_bytes = 0, mailbox_bytes = 0;
  char* mailbox = nullptr;
  char* peers[kMaxRanks] = {};
  bool opened[kMaxRanks] = {};
  uint32_t* state = nullptr;  // device: [0] calls completed, [1] workgroups done in this call, [2] status
  uint32_t* host_flag = nullptr;      // pinned, mapped: the status word as the host reads it
  uint32_t* host_flag_dev = nullptr;  // the same word as the kernels address it
  long long timeout_ticks = 300000LL * 100000;  // wall_clock64() ticks at 100 MHz
  bool connected = false;
};

struct ArArgs {
  const uint4* in;
  uint4* out;
  char* mbox[kMaxRanks];
  uint32_t* state;
  uint32_t* host_flag;
  long long units;       // 16-byte units in the vector
  long long region_bytes, half_bytes;
  long long timeout_ticks;
  int rank;
};

int64_t round_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

int64_t one_stage_cap(int world, int64_t max_bytes) {
  if (world == 2) return max_bytes;
  const int64_t lim = world <= 4 ? 512 * 1024 : 256 * 1024;
  return max_bytes < lim ? max_bytes : lim;
}

void layout(int world, int64_t max_bytes, int64_t* half, int64_t* region) {
  *half = round_up(max_bytes + 4096, 256);  // world slots of floor(units / world) + world units each
  const int64_t one = round_up(world * one_stage_cap(world, max_bytes), 256);
  *region = 2 * *half > one ? 2 * *half : one;
}

bool one_stage(int world, int64_t bytes) {
  return world == 2 || (world <= 4 && bytes < 512 * 1024) || bytes < 256 * 1024;
}

// ---------------------------------------------------------------- device side

__device__ inline uint32_t* flag_ptr(char* mbox, int half, int stage, int src, int block) {
  return reinterpret_cast<uint32_t*>(mbox) + ((half * 2 + stage) * kMaxRanks + src) * kMaxBlocks + block;
}

__device__ inline uint4* data_ptr(char* mbox, int half, long long region_bytes) {
  return reinterpret_cast<uint4*>(mbox + kFlagBytes + half * region_bytes);
}

template <typename T> struct Elem;

template <> struct Elem<float> {
  static constexpr int N = 4;
  __device__ static void unpack(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x), f[1] = __uint_as_float(v.y), f[2] = __uint_as_float(v.z), f[3] = __uint_as_float(v.w);
  }
  __device__ static uint4 pack(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
};

template <> struct Elem<__half> {
  static constexpr int N = 8;
  __device__ static void unpack(const uint4& v, float* f) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __half2float(__ushort_as_half(static_cast<unsigned short>(w[i] & 0xffffu)));
      f[2 * i + 1] = __half2float(__ushort_as_half(static_cast<unsigned short>(w[i] >> 16)));
    }
  }
  __device__ static uint4 pack(const float* f) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      w[i] = static_cast<uint32_t>(__half_as_ushort(__float2half_rn(f[2 * i]))) |
             static_cast<uint32_t>(__half_as_ushort(__float2half_rn(f[2 * i + 1]))) << 16;
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
};

struct bf16_tag {};

template <> struct Elem<bf16_tag> {
  static constexpr int N = 8;
  __device__ static void unpack(const uint4& v, float* f) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(w[i] << 16);
      f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  // round to nearest even; NaN stays NaN
  __device__ static uint32_t round_bf16(float x) {
    uint32_t u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
  }
  __device__ static uint4 pack(const float* f) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = round_bf16(f[2 * i]) | round_bf16(f[2 * i + 1]) << 16;
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
};

// Makes this workgroup's mailbox writes visible, tells every peer's same-numbered workgroup, and
// waits for theirs.  Threads 0..NG-1 (one per peer) do the flag traffic.
template <int NG>
__device__ inline void signal_and_wait(const ArArgs& a, int half, int stage, uint32_t e) {
  __threadfence_system();
  __syncthreads();
  const int t = threadIdx.x;
  if (t < NG && t != a.rank) {
    __hip_atomic_store(flag_ptr(a.mbox[t], half, stage, a.rank, blockIdx.x), e, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
    uint32_t* mine = flag_ptr(a.mbox[a.rank], half, stage, t, blockIdx.x);
    const long long t0 = wall_clock64();
    unsigned long long spins = 0;
    while (static_cast<int>(__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
      ++spins;
      if (spins > kSpinCap || ((spins & 63u) == 0 && wall_clock64() - t0 > a.timeout_ticks)) {
        a.state[2] = 1;  // gave up: sticky, read by p2p_ar_status()
        __hip_atomic_store(a.host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // and by p2p_ar_poll()
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: nothing read earlier may stand in for the mailbox
}

// The last workgroup of a call to finish advances the call counter.  Every workgroup read the
// counter at its start, and the next call's kernel starts after this one has ended.
__device__ inline void finish_call(const ArArgs& a, uint32_t e) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t done = atomicAdd(&a.state[1], 1u);
    if (done == gridDim.x - 1) {
      a.state[1] = 0;
      a.state[0] = e;
    }
  }
}

template <typename T, int NG>
__global__ __launch_bounds__(kThreads) void all_reduce_one_stage(ArArgs a) {
  using E = Elem<T>;
  const uint32_t e = a.state[0] + 1;
  const int half = e & 1;
  const long long u0 = a.units * blockIdx.x / gridDim.x, u1 = a.units * (blockIdx.x + 1) / gridDim.x;

  for (long long i = u0 + threadIdx.x; i < u1; i += kThreads) {
    const uint4 v = a.in[i];
#pragma unroll
    for (int t = 0; t < NG; ++t)
      if (t != a.rank) data_ptr(a.mbox[t], half, a.region_bytes)[a.rank * a.units + i] = v;
  }
  signal_and_wait<NG>(a, half, 0, e);

  const uint4* mine = data_ptr(a.mbox[a.rank], half, a.region_bytes);
  for (long long i = u0 + threadIdx.x; i < u1; i += kThreads) {
    float acc[E::N], f[E::N];
#pragma unroll
    for (int s = 0; s < NG; ++s) {
      const uint4 v = s == a.rank ? a.in[i] : mine[s * a.units + i];
      if (s == 0) {
        E::unpack(v, acc);
      } else {
        E::unpack(v, f);
#pragma unroll
        for (int j = 0; j < E::N; ++j) acc[j] += f[j];
      }
    }
    a.out[i] = E::pack(acc);
  }
  finish_call(a, e);
}

template <typename T, int NG>
__global__ __launch_bounds__(kThreads) void all_reduce_two_stage(ArArgs a) {
  using E = Elem<T>;
  const uint32_t e = a.state[0] + 1;
  const int half = e & 1;
  const int B = gridDim.x, b = blockIdx.x;
  const long long part = a.units / NG, slot_units = part + NG;
  auto part_len = [&](int p) { return p == NG - 1 ? a.units - part * (NG - 1) : part; };
  const long long stage2 = a.half_bytes / 16;  // unit offset of the stage-2 slots

  // stage 1: send my piece of part p to rank p
#pragma unroll
  for (int p = 0; p < NG; ++p) {
    if (p == a.rank) continue;
    const long long len = part_len(p), c0 = len * b / B, c1 = len * (b + 1) / B;
    uint4* dst = data_ptr(a.mbox[p], half, a.region_bytes) + a.rank * slot_units;
    for (long long i = c0 + threadIdx.x; i < c1; i += kThreads) dst[i] = a.in[p * part + i];
  }
  signal_and_wait<NG>(a, half, 0, e);

  // reduce my part in the order rank, rank + 1, ..; keep it and send it to everyone
  {
    const long long len = part_len(a.rank), c0 = len * b / B, c1 = len * (b + 1) / B;
    const uint4* mine = data_ptr(a.mbox[a.rank], half, a.region_bytes);
    for (long long i = c0 + threadIdx.x; i < c1; i += kThreads) {
      float acc[E::N], f[E::N];
      E::unpack(a.in[a.rank * part + i], acc);
#pragma unroll
      for (int k = 1; k < NG; ++k) {
        const int s = (a.rank + k) % NG;
        E::unpack(mine[s * slot_units + i], f);
#pragma unroll
        for (int j = 0; j < E::N; ++j) acc[j] += f[j];
      }
      const uint4 r = E::pack(acc);
      a.out[a.rank * part + i] = r;
#pragma unroll
      for (int t = 0; t < NG; ++t)
        if (t != a.rank) data_ptr(a.mbox[t], half, a.region_bytes)[stage2 + a.rank * slot_units + i] = r;
    }
  }
  signal_and_wait<NG>(a, half, 1, e);

  // stage 2: collect the other parts
  const uint4* mine = data_ptr(a.mbox[a.rank], half, a.region_bytes) + stage2;
#pragma unroll
  for (int p = 0; p < NG; ++p) {
    if (p == a.rank) continue;
    const long long len = part_len(p), c0 = len * b / B, c1 = len * (b + 1) / B;
    for (long long i = c0 + threadIdx.x; i < c1; i += kThreads) a.out[p * part + i] = mine[p * slot_units + i];
  }
  finish_call(a, e);
}

// ---------------------------------------------------------------- host side

int hip_fail(hipError_t err) {
  g_last_hip_error = static_cast<int>(err);
  (void)hipGetLastError();
  return P2P_AR_E_HIP;
}

#define HIP_TRY(expr)                                 \
  do {                                                \
    const hipError_t err__ = (expr);                  \
    if (err__ != hipSuccess) return hip_fail(err__);  \
  } while (0)

template <typename T, int NG>
int launch(const ArArgs& a, bool one, int blocks, hipStream_t s) {
  if (one)
    hipLaunchKernelGGL((all_reduce_one_stage<T, NG>), dim3(blocks), dim3(kThreads), 0, s, a);
  else
    hipLaunchKernelGGL((all_reduce_two_stage<T, NG>), dim3(blocks), dim3(kThreads), 0, s, a);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? P2P_AR_OK : hip_fail(err);
}

template <typename T>
int launch_world(int world, const ArArgs& a, bool one, int blocks, hipStream_t s) {
  switch (world) {
    case 2: return launch<T, 2>(a, one, blocks, s);
    case 4: return launch<T, 4>(a, one, blocks, s);
    case 6: return launch<T, 6>(a, one, blocks, s);
    case 8: return launch<T, 8>(a, one, blocks, s);
  }
  return P2P_AR_E_WORLD;
}

bool world_ok(int world) { return world == 2 || world == 4 || world == 6 || world == 8; }

}  // namespace

extern "C" {

const char* p2p_ar_strerror(int code) {
  switch (code) {
    case P2P_AR_OK: return "ok";
    case P2P_AR_E_WORLD: return "custom allreduce only supports num gpus in (2,4,6,8) and a rank inside the group";
    case P2P_AR_E_NULL: return "a required pointer is NULL";
    case P2P_AR_E_SIZE: return "byte size must be a positive multiple of 16 and at most the context's max size";
    case P2P_AR_E_ALIGN: return "input and output must be 16-byte aligned";
    case P2P_AR_E_DTYPE: return "custom allreduce only supports float32, float16 and bfloat16";
    case P2P_AR_E_HIP: return "a HIP runtime call failed";
    case P2P_AR_E_TIMEOUT: return "an earlier all-reduce of this context gave up waiting for a peer; its result is invalid";
    case P2P_AR_E_STATE: return "context is not connected to its peers (or already is), or its GPU is not the current device";
  }
  return "unknown error code";
}

int p2p_ar_last_hip_error(void) { return g_last_hip_error; }
int p2p_ar_abi_version(void) { return P2P_AR_ABI_VERSION; }

int p2p_ar_should(int64_t bytes, int64_t max_bytes, int world, int full_mesh) {
  if (bytes <= 0 || bytes % 16 != 0) return 0;
  if (world == 2 || full_mesh) return bytes <= max_bytes;
  return world <= 4 && bytes <= 512 * 1024 && bytes <= max_bytes;
}

int64_t p2p_ar_mailbox_bytes(int world, int64_t max_bytes) {
  if (!world_ok(world) || max_bytes <= 0) return 0;
  int64_t half, region;
  layout(world, max_bytes, &half, &region);
  return kFlagBytes + 2 * region;
}

int p2p_ar_create(int rank, int world, int64_t max_bytes, void** ctx_out) {
  if (!ctx_out) return P2P_AR_E_NULL;
  *ctx_out = nullptr;
  if (!world_ok(world) || rank < 0 || rank >= world) return P2P_AR_E_WORLD;
  if (max_bytes <= 0 || max_bytes % 16 != 0 || max_bytes > (int64_t(1) << 30)) return P2P_AR_E_SIZE;
  Ctx* c = new (std::nothrow) Ctx;
  if (!c) return P2P_AR_E_NULL;
  c->rank = rank, c->world = world, c->max_bytes = max_bytes;
  layout(world, max_bytes, &c->half_bytes, &c->region_bytes);
  c->mailbox_bytes = kFlagBytes + 2 * c->region_bytes;
  hipError_t err = hipGetDevice(&c->device);
  void* p = nullptr;
  if (err == hipSuccess) err = hipExtMallocWithFlags(&p, c->mailbox_bytes, hipDeviceMallocUncached);
  c->mailbox = static_cast<char*>(p);
  if (err == hipSuccess) err = hipMemset(c->mailbox, 0, c->mailbox_bytes);
  if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&c->state), 64);
  if (err == hipSuccess) err = hipMemset(c->state, 0, 64);
  if (err == hipSuccess) err = hipHostMalloc(reinterpret_cast<void**>(&c->host_flag), 64, hipHostMallocMapped);
  if (err == hipSuccess) {
    *c->host_flag = 0;
    err = hipHostGetDevicePointer(reinterpret_cast<void**>(&c->host_flag_dev), c->host_flag, 0);
  }
  if (err == hipSuccess) err = hipDeviceSynchronize();
  if (err != hipSuccess) {
    if (c->mailbox) (void)hipFree(c->mailbox);
    if (c->state) (void)hipFree(c->state);
    if (c->host_flag) (void)hipHostFree(c->host_flag);
    delete c;
    return hip_fail(err);
  }
  c->peers[rank] = c->mailbox;
  *ctx_out = c;
  return P2P_AR_OK;
}

int p2p_ar_ipc_handle(void* ctx, void* handle) {
  if (!ctx || !handle) return P2P_AR_E_NULL;
  static_assert(sizeof(hipIpcMemHandle_t) == P2P_AR_IPC_HANDLE_BYTES, "IPC handle size");
  Ctx* c = static_cast<Ctx*>(ctx);
  hipIpcMemHandle_t h;
  HIP_TRY(hipIpcGetMemHandle(&h, c->mailbox));
  std::memcpy(handle, &h, sizeof(h));
  return P2P_AR_OK;
}

int p2p_ar_connect(void* ctx, const void* handles) {
  if (!ctx || !handles) return P2P_AR_E_NULL;
  Ctx* c = static_cast<Ctx*>(ctx);
  if (c->connected) return P2P_AR_E_STATE;
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank) continue;
    hipIpcMemHandle_t h;
    std::memcpy(&h, static_cast<const char*>(handles) + r * sizeof(h), sizeof(h));
    void* p = nullptr;
    HIP_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    c->peers[r] = static_cast<char*>(p);
    c->opened[r] = true;
  }
  c->connected = true;
  return P2P_AR_OK;
}

void* p2p_ar_mailbox_ptr(void* ctx) { return ctx ? static_cast<Ctx*>(ctx)->mailbox : nullptr; }

int p2p_ar_connect_ptrs(void* ctx, void* const* mailboxes) {
  if (!ctx || !mailboxes) return P2P_AR_E_NULL;
  Ctx* c = static_cast<Ctx*>(ctx);
  if (c->connected) return P2P_AR_E_STATE;
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank) continue;
    if (!mailboxes[r]) return P2P_AR_E_NULL;
    c->peers[r] = static_cast<char*>(mailboxes[r]);
  }
  c->connected = true;
  return P2P_AR_OK;
}

int p2p_ar_all_reduce(void* ctx, const void* in, void* out, int64_t numel, int dtype, void* stream) {
  if (!ctx || !in || !out) return P2P_AR_E_NULL;
  Ctx* c = static_cast<Ctx*>(ctx);
  if (!c->connected) return P2P_AR_E_STATE;
  if (dtype != P2P_AR_F32 && dtype != P2P_AR_F16 && dtype != P2P_AR_BF16) return P2P_AR_E_DTYPE;
  const int64_t esize = dtype == P2P_AR_F32 ? 4 : 2;
  if (numel <= 0 || numel > c->max_bytes / esize) return P2P_AR_E_SIZE;
  const int64_t bytes = numel * esize;
  if (bytes % 16 != 0) return P2P_AR_E_SIZE;
  if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) return P2P_AR_E_ALIGN;
  int dev = -1;
  HIP_TRY(hipGetDevice(&dev));
  if (dev != c->device) return P2P_AR_E_STATE;  // the mailbox and the call counter live on the context's GPU
  if (*static_cast<volatile uint32_t*>(c->host_flag)) return P2P_AR_E_TIMEOUT;

  ArArgs a;
  a.in = static_cast<const uint4*>(in);
  a.out = static_cast<uint4*>(out);
  for (int r = 0; r < kMaxRanks; ++r) a.mbox[r] = c->peers[r];
  a.state = c->state;
  a.host_flag = c->host_flag_dev;
  a.units = bytes / 16;
  a.region_bytes = c->region_bytes;
  a.half_bytes = c->half_bytes;
  a.timeout_ticks = c->timeout_ticks;
  a.rank = c->rank;

  // every rank derives the same grid from (world, units): the flags pair up same-numbered workgroups
  const bool one = one_stage(c->world, bytes) || a.units < 2 * c->world;
  const int64_t per_rank = one ? a.units : a.units / c->world;
  int64_t blocks = (per_rank + kThreads - 1) / kThreads;
  const int64_t cap = one ? kOneStageBlocks : kMaxBlocks;
  blocks = blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case P2P_AR_F32: return launch_world<float>(c->world, a, one, static_cast<int>(blocks), s);
    case P2P_AR_F16: return launch_world<__half>(c->world, a, one, static_cast<int>(blocks), s);
    default: return launch_world<bf16_tag>(c->world, a, one, static_cast<int>(blocks), s);
  }
}

int p2p_ar_set_timeout_ms(void* ctx, int ms) {
  if (!ctx) return P2P_AR_E_NULL;
  if (ms <= 0) return P2P_AR_E_SIZE;
  static_cast<Ctx*>(ctx)->timeout_ticks = static_cast<long long>(ms) * 100000;
  return P2P_AR_OK;
}

int p2p_ar_status(void* ctx, int* timed_out, int clear) {
  if (!ctx || !timed_out) return P2P_AR_E_NULL;
  Ctx* c = static_cast<Ctx*>(ctx);
  uint32_t v = 0;
  HIP_TRY(hipMemcpy(&v, c->state + 2, 4, hipMemcpyDeviceToHost));
  *timed_out = v != 0;
  if (*static_cast<volatile uint32_t*>(c->host_flag)) *timed_out = 1;
  if (clear) {
    if (v) HIP_TRY(hipMemset(c->state + 2, 0, 4));
    *static_cast<volatile uint32_t*>(c->host_flag) = 0;
  }
  return P2P_AR_OK;
}

int p2p_ar_poll(void* ctx, int* timed_out) {
  if (!ctx || !timed_out) return P2P_AR_E_NULL;
  *timed_out = *static_cast<volatile uint32_t*>(static_cast<Ctx*>(ctx)->host_flag) != 0;
  return P2P_AR_OK;
}

int p2p_ar_destroy(void* ctx) {
  if (!ctx) return P2P_AR_E_NULL;
  Ctx* c = static_cast<Ctx*>(ctx);
  int rc = P2P_AR_OK;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < c->world; ++r)
    if (c->opened[r] && hipIpcCloseMemHandle(c->peers[r]) != hipSuccess) rc = P2P_AR_E_HIP;
  if (c->mailbox && hipFree(c->mailbox) != hipSuccess) rc = P2P_AR_E_HIP;
  if (c->state && hipFree(c->state) != hipSuccess) rc = P2P_AR_E_HIP;
  if (c->host_flag && hipHostFree(c->host_flag) != hipSuccess) rc = P2P_AR_E_HIP;
  delete c;
  return rc;
}

}  // extern "C"
