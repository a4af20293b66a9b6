// A host program with no PyTorch in it: drives libp2p_ar.so through include/p2p_allreduce.h with one
// process per rank (fork before any HIP call), IPC handles exchanged over pipes, plain hipMalloc
// buffers, and checks every rank's result bit for bit against sums taken on the CPU in the order the
// header states.  All ranks use GPU 0, which is what a one-GPU machine can offer.
//
//   ./p2p_abi_check [world=4]
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "p2p_allreduce.h"

#define HIP_OK(expr)                                                                   \
  do {                                                                                 \
    hipError_t e__ = (expr);                                                           \
    if (e__ != hipSuccess) {                                                           \
      std::fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e__));                 \
      return 2;                                                                        \
    }                                                                                  \
  } while (0)
#define AR_OK(expr)                                                                    \
  do {                                                                                 \
    int c__ = (expr);                                                                  \
    if (c__ != 0) {                                                                    \
      std::fprintf(stderr, "%s: %s\n", #expr, p2p_ar_strerror(c__));                   \
      return 3;                                                                        \
    }                                                                                  \
  } while (0)

static bool read_all(int fd, void* buf, size_t n) {
  char* p = static_cast<char*>(buf);
  while (n) {
    ssize_t r = read(fd, p, n);
    if (r <= 0) return false;
    p += r, n -= static_cast<size_t>(r);
  }
  return true;
}

static bool write_all(int fd, const void* buf, size_t n) {
  const char* p = static_cast<const char*>(buf);
  while (n) {
    ssize_t r = write(fd, p, n);
    if (r <= 0) return false;
    p += r, n -= static_cast<size_t>(r);
  }
  return true;
}

// rank r's element i of case c: reproducible everywhere, not exactly representable sums
static float value(int c, int r, long long i) {
  uint32_t h = static_cast<uint32_t>(i * 2654435761u) ^ (static_cast<uint32_t>(r + 1) * 40503u) ^ (c * 977u);
  h ^= h >> 13, h *= 0x5bd1e995u, h ^= h >> 15;
  return (static_cast<int>(h % 20001u) - 10000) * 1.0e-3f + r * 0.125f;
}

static int run_rank(int rank, int world, int to_parent, int from_parent) {
  HIP_OK(hipSetDevice(0));
  const int64_t max_bytes = 4 << 20;
  void* ctx = nullptr;
  AR_OK(p2p_ar_create(rank, world, max_bytes, &ctx));
  char mine[P2P_AR_IPC_HANDLE_BYTES];
  AR_OK(p2p_ar_ipc_handle(ctx, mine));
  if (!write_all(to_parent, mine, sizeof(mine))) return 4;
  std::vector<char> all(static_cast<size_t>(world) * P2P_AR_IPC_HANDLE_BYTES);
  if (!read_all(from_parent, all.data(), all.size())) return 4;
  AR_OK(p2p_ar_connect(ctx, all.data()));
  AR_OK(p2p_ar_set_timeout_ms(ctx, 20000));

  // fp32 sizes on both sides of the one-stage / two-stage switch (256 KiB from 6 ranks, 512 KiB up to 4)
  const long long sizes[] = {4, 4096, 60 * 1024, 64 * 1024, 128 * 1024, 1000 * 1000, 1 << 20};
  float *d_in = nullptr, *d_out = nullptr;
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&d_in), max_bytes));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&d_out), max_bytes));
  std::vector<float> h(1 << 20), got(1 << 20);
  int bad = 0, c = 0;
  for (long long n : sizes) {
    ++c;
    for (long long i = 0; i < n; ++i) h[i] = value(c, rank, i);
    HIP_OK(hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice));
    AR_OK(p2p_ar_all_reduce(ctx, d_in, d_out, n, P2P_AR_F32, nullptr));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(got.data(), d_out, n * 4, hipMemcpyDeviceToHost));
    const long long bytes = n * 4, units = bytes / 16, part = units / world;
    const bool one = world == 2 || (world <= 4 && bytes < 512 * 1024) || bytes < 256 * 1024 || units < 2 * world;
    long long wrong = 0;
    for (long long i = 0; i < n; ++i) {
      int first = 0;
      if (!one) {
        long long p = (i / 4) / part;
        first = static_cast<int>(p >= world ? world - 1 : p);
      }
      volatile float acc = value(c, first, i);
      for (int k = 1; k < world; ++k) acc = acc + value(c, (first + k) % world, i);
      float want = acc;
      if (std::memcmp(&want, &got[i], 4) != 0) ++wrong;
    }
    std::printf("rank %d  n=%-8lld %s  %s\n", rank, n, one ? "one-stage" : "two-stage", wrong ? "MISMATCH" : "ok");
    bad += wrong != 0;
  }
  // argument errors
  bad += p2p_ar_all_reduce(ctx, d_in, d_out, 6, P2P_AR_F32, nullptr) != P2P_AR_E_SIZE;
  bad += p2p_ar_all_reduce(ctx, d_in, d_out, 8, 9, nullptr) != P2P_AR_E_DTYPE;
  bad += p2p_ar_all_reduce(ctx, reinterpret_cast<char*>(d_in) + 4, d_out, 8, P2P_AR_F32, nullptr) != P2P_AR_E_ALIGN;
  int timed_out = 0;
  AR_OK(p2p_ar_status(ctx, &timed_out, 0));
  bad += timed_out;
  // nobody frees a mailbox a peer may still write to: tell the parent, wait for its go
  char done = 1, go = 0;
  if (!write_all(to_parent, &done, 1) || !read_all(from_parent, &go, 1)) return 4;
  HIP_OK(hipFree(d_in));
  HIP_OK(hipFree(d_out));
  AR_OK(p2p_ar_destroy(ctx));
  return bad ? 1 : 0;
}

int main(int argc, char** argv) {
  const int world = argc > 1 ? std::atoi(argv[1]) : 4;
  if (world != 2 && world != 4 && world != 6 && world != 8) {
    std::fprintf(stderr, "world must be 2, 4, 6 or 8\n");
    return 2;
  }
  std::vector<int> up(world), down(world);
  std::vector<pid_t> pids(world);
  for (int r = 0; r < world; ++r) {
    int a[2], b[2];
    if (pipe(a) || pipe(b)) return 2;
    pid_t pid = fork();  // before any HIP call in this process
    if (pid < 0) return 2;
    if (pid == 0) {
      close(a[0]), close(b[1]);
      for (int q = 0; q < r; ++q) close(up[q]), close(down[q]);
      const int rc = run_rank(r, world, a[1], b[0]);
      std::fflush(stdout);
      _exit(rc);
    }
    close(a[1]), close(b[0]);
    up[r] = a[0], down[r] = b[1], pids[r] = pid;
  }
  std::vector<char> all(static_cast<size_t>(world) * P2P_AR_IPC_HANDLE_BYTES);
  bool ok = true;
  for (int r = 0; r < world; ++r) ok = ok && read_all(up[r], all.data() + r * P2P_AR_IPC_HANDLE_BYTES, P2P_AR_IPC_HANDLE_BYTES);
  for (int r = 0; r < world && ok; ++r) ok = write_all(down[r], all.data(), all.size());
  char done = 0, go = 1;
  for (int r = 0; r < world && ok; ++r) ok = read_all(up[r], &done, 1);
  for (int r = 0; r < world; ++r) (void)write_all(down[r], &go, 1);
  int failures = ok ? 0 : 1;
  for (int r = 0; r < world; ++r) {
    int st = 0;
    waitpid(pids[r], &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
      std::fprintf(stderr, "rank %d failed (status %d)\n", r, st);
      ++failures;
    }
  }
  std::printf(failures ? "FAILED\n" : "ALL OK\n");
  return failures ? 1 : 0;
}
