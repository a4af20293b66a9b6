/*
 * p2p_allreduce.h - C ABI of libp2p_ar.so: latency-oriented all-reduce(sum) between the GPUs of
 * one node for AMD MI355X (gfx950), for the tensor-parallel row-split linear layer's partial sums.
 *
 * Every rank owns one "mailbox": device memory allocated uncached (fine-grained) by this library
 * and mapped into every peer process through a HIP IPC handle.  An all-reduce is ONE kernel per
 * rank that WRITES its contribution straight into the peers' mailboxes over xGMI (posted writes,
 * no read round trip), raises a per-workgroup flag in each peer's mailbox, waits for the peers'
 * flags in its own, and sums what arrived from local memory.  Inputs and outputs are ordinary
 * device buffers that need no registration; the call counter that selects the mailbox half and
 * the flag value lives in device memory and is advanced by the kernel itself, so a captured HIP
 * graph can be replayed any number of times, interleaved with eager calls.
 *
 * Replaces the reference's `custom_ar` extension, which the reference does not build on ROCm
 * (csrc/pybind.cpp:99-113): csrc/custom_all_reduce.cu:12-148 (host side),
 * csrc/custom_all_reduce.cuh:144-290 (kernels), :486-548 (algorithm choice).
 *
 * Arithmetic (identical on every rank, so all ranks hold bitwise equal results, and identical to
 * the reference's, custom_all_reduce.cuh:204-212,221-229,248-264): elements are converted to
 * fp32, added one rank after another, and rounded once to the element type.  One-stage: ranks in
 * the order 0, 1, ..  Two-stage: the vector is cut into `world` parts of floor(units / world)
 * 16-byte units, the last part takes the remainder; part p is summed by rank p in the order
 * p, p+1, .. (mod world), rounded to the element type and sent to everyone.
 * One-stage is used when world == 2, or bytes < 512 KiB with world <= 4, or bytes < 256 KiB
 * (custom_all_reduce.cuh:520-528, the fully connected case: xGMI is a full mesh).
 *
 * Collective contract: every rank of the group must make the same sequence of
 * p2p_ar_all_reduce() calls (same element count and type).  Calls of one context must be
 * stream-ordered with respect to each other.  A rank that waits longer than the context's
 * timeout for a peer stops waiting, leaves an unspecified result and raises the context's
 * sticky status flag; it never spins without bound.  The flag also lives in host memory the
 * kernel writes directly, so the host sees it without synchronising: every later
 * p2p_ar_all_reduce() of the context fails with P2P_AR_E_TIMEOUT until the flag is cleared
 * (p2p_ar_status with clear != 0), and p2p_ar_poll() reads it at any time.  (The reference's
 * kernels spin until the peer arrives, custom_all_reduce.cuh:144-178: a late rank delays the
 * others; here it is reported once the timeout, 5 minutes unless set, has passed.)
 *
 * Return value: 0 on success, a negative P2P_AR_E* code otherwise.
 */
#ifndef P2P_ALLREDUCE_H_
#define P2P_ALLREDUCE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P2P_AR_ABI_VERSION 2
#define P2P_AR_IPC_HANDLE_BYTES 64 /* sizeof(hipIpcMemHandle_t) */

enum {
  P2P_AR_OK = 0,
  P2P_AR_E_WORLD = -1,     /* world size not in {2, 4, 6, 8} or rank outside it (custom_all_reduce.cu:17-25) */
  P2P_AR_E_NULL = -2,      /* a required pointer is NULL                                                      */
  P2P_AR_E_SIZE = -3,      /* byte size not a positive multiple of 16 (custom_all_reduce.cuh:489-494) or > max */
  P2P_AR_E_ALIGN = -4,     /* input or output not 16-byte aligned                                             */
  P2P_AR_E_DTYPE = -5,     /* element type not f32 / f16 / bf16 (custom_all_reduce.cu:95-97)                  */
  P2P_AR_E_HIP = -6,       /* a HIP runtime call failed; see p2p_ar_last_hip_error()                          */
  P2P_AR_E_STATE = -7,     /* context not connected yet, connected twice, or its GPU is not the current device */
  P2P_AR_E_TIMEOUT = -8    /* an earlier kernel of this context gave up waiting for a peer: its result is invalid */
};

enum { P2P_AR_F32 = 0, P2P_AR_F16 = 1, P2P_AR_BF16 = 2 };

const char* p2p_ar_strerror(int code);
int p2p_ar_last_hip_error(void);
int p2p_ar_abi_version(void);

/*
 * Whether an input of `bytes` bytes takes this path: a multiple of 16 bytes and at most
 * `max_bytes`.  Replaces should_custom_ar(), custom_all_reduce.cu:58-68, for the fully connected
 * case; `full_mesh == 0` with more than 2 ranks keeps the reference's limit for a node without
 * direct links (world <= 4 and at most 512 KiB).  The reference's contiguity test is the caller's.
 */
int p2p_ar_should(int64_t bytes, int64_t max_bytes, int world, int full_mesh);

/* Bytes of mailbox memory a context for (world, max_bytes) allocates on its GPU. */
int64_t p2p_ar_mailbox_bytes(int world, int64_t max_bytes);

/*
 * Creates this rank's context on the current HIP device: allocates and zeroes the mailbox and
 * the call counter (the only allocations this library makes), synchronises the device.
 * Replaces init_custom_ar(), custom_all_reduce.cu:12-34, with the buffers owned here instead of
 * by the caller.  `*ctx` receives an opaque host pointer.
 */
int p2p_ar_create(int rank, int world, int64_t max_bytes, void** ctx);

/* Copies the mailbox's IPC handle (P2P_AR_IPC_HANDLE_BYTES bytes) to the host buffer `handle`. */
int p2p_ar_ipc_handle(void* ctx, void* handle);

/*
 * Maps the peers' mailboxes.  `handles`: host buffer of world x P2P_AR_IPC_HANDLE_BYTES bytes,
 * entry r = what rank r's p2p_ar_ipc_handle() returned (this rank's own entry is ignored).
 * The caller exchanges the handles (e.g. with an all-gather over its control-plane group), as
 * custom_all_reduce.py:173-190 does.
 */
int p2p_ar_connect(void* ctx, const void* handles);

/*
 * Same, for ranks that live in ONE process (several GPUs with peer access enabled, or several
 * contexts on one GPU): `mailboxes` is a host array of `world` device pointers, entry r = rank
 * r's p2p_ar_mailbox_ptr().
 */
int p2p_ar_connect_ptrs(void* ctx, void* const* mailboxes);
void* p2p_ar_mailbox_ptr(void* ctx);

/*
 * out = sum over ranks of in, `numel` elements of `dtype` (P2P_AR_F32 / F16 / BF16); enqueues one
 * kernel on `stream` (hipStream_t, NULL = default stream) and returns; allocates nothing; may be
 * captured in a HIP graph.  `out` may equal `in`.  Replaces all_reduce_reg() and
 * all_reduce_unreg(), custom_all_reduce.cu:101-122 (no staging copy is needed: the kernel reads
 * `in` wherever it is).
 */
int p2p_ar_all_reduce(void* ctx, const void* in, void* out, int64_t numel, int dtype, void* stream);

/* How long a kernel waits for a peer before it gives up (default 300000 ms). */
int p2p_ar_set_timeout_ms(void* ctx, int ms);

/*
 * Reads the host copy of the sticky status flag: no device call, no synchronisation, may be called
 * at any time (also while kernels of the context run or a stream is being captured).  *timed_out = 1
 * once a kernel that gave up has written the flag; a kernel still waiting has not.
 */
int p2p_ar_poll(void* ctx, int* timed_out);

/*
 * Synchronous read of the sticky status flag: *timed_out = 1 if any kernel of this context gave
 * up waiting since the last call with `clear` != 0.  The caller synchronises the stream first.
 */
int p2p_ar_status(void* ctx, int* timed_out, int clear);

/* Unmaps the peers' mailboxes and frees this rank's.  Replaces dispose(), custom_all_reduce.cu:124-127. */
int p2p_ar_destroy(void* ctx);

#ifdef __cplusplus
}
#endif
#endif /* P2P_ALLREDUCE_H_ */
