"""CPU oracle for the AWQ W4A16 GEMM path (numpy).

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package imports this module:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it,
and there only as the checker.

PARITY UNPINNED: the reference holds no test, golden vector or fixture for the
AWQ kernels, and its CUDA/PTX sources cannot be built in this image (no nvcc,
inline PTX).  This file is therefore a restatement of the reference's
arithmetic, read from its sources, cross-checked against a bit-level emulation
of the PTX sequence (`s4_to_fp16x2_bits` below) and against an independent C
restatement (oracle/awq_oracle.c).

What is restated, with the reference lines each function follows
(paths relative to the reference checkout):

* nibble order of a packed word      csrc/quantization/awq/dequantize.cuh:17-84
* per-element dequantisation         csrc/quantization/awq/gemm_kernels.cu:128-130,147-160
* dequantize-to-dense op             csrc/quantization/awq/gemm_kernels.cu:274-381
* split-K GEMM and its final sum     csrc/quantization/awq/gemm_kernels.cu:112-115,262-270,389-446
* host-side shape checks             csrc/quantization/awq/gemm_kernels.cu:412-419
"""
from __future__ import annotations

import numpy as np

# Output column j (0..7) of one packed 32-bit word comes from nibble
# AWQ_ORDER[j] (nibble p = bits 4p..4p+3).  dequantize.cuh:40-56 extracts
# nibbles {0,4}, {1,5}, {2,6}, {3,7} into four half2 registers and
# gemm_kernels.cu:168 stores them as 8 consecutive columns.
AWQ_ORDER = (0, 4, 1, 5, 2, 6, 3, 7)
PACK = 8


# --------------------------------------------------------------------------
# packing helpers
# --------------------------------------------------------------------------
def unpack_words(words: np.ndarray) -> np.ndarray:
    """int32/uint32 [..., C] -> uint8 [..., 8*C]; column 8c+j = nibble AWQ_ORDER[j]."""
    w = np.ascontiguousarray(words).view(np.uint32)
    out = np.empty(w.shape + (PACK,), dtype=np.uint8)
    for j, p in enumerate(AWQ_ORDER):
        out[..., j] = (w >> np.uint32(4 * p)) & np.uint32(0xF)
    return out.reshape(w.shape[:-1] + (w.shape[-1] * PACK,))


def pack_words(vals: np.ndarray) -> np.ndarray:
    """uint8 [..., 8*C] with values 0..15 -> int32 [..., C] (inverse of unpack_words)."""
    v = np.asarray(vals)
    if v.shape[-1] % PACK:
        raise ValueError("last dimension must be a multiple of 8")
    if v.size and (v.min() < 0 or v.max() > 15):
        raise ValueError("values must be 4-bit")
    v = v.reshape(v.shape[:-1] + (v.shape[-1] // PACK, PACK)).astype(np.uint32)
    w = np.zeros(v.shape[:-1], dtype=np.uint32)
    for j, p in enumerate(AWQ_ORDER):
        w |= v[..., j] << np.uint32(4 * p)
    return w.view(np.int32)


# --------------------------------------------------------------------------
# bit-level emulation of dequantize_s4_to_fp16x2 (dequantize.cuh:17-84)
# --------------------------------------------------------------------------
def _h2(bits32: np.ndarray) -> np.ndarray:
    """uint32 [...] -> float16 [..., 2] (low half first)."""
    b = np.ascontiguousarray(bits32, dtype=np.uint32)
    return b.view(np.uint16).reshape(b.shape + (2,)).view(np.float16)


def s4_to_fp16x2_bits(words: np.ndarray) -> np.ndarray:
    """Emulates the PTX sequence instruction by instruction.

    uint32 [...] -> float16 [..., 8]: the four half2 registers h[0..3] in order.
    lop3 with immLut (0xf0 & 0xcc) | 0xaa computes (a & b) | c; sub.f16x2 and
    fma.rn.f16x2 round once per lane.
    """
    i4s = np.ascontiguousarray(words).view(np.uint32)
    top = i4s >> np.uint32(8)
    BOTTOM, TOP, MAGIC = np.uint32(0x000F000F), np.uint32(0x00F000F0), np.uint32(0x64006400)
    h0 = _h2((i4s & BOTTOM) | MAGIC)
    h1 = _h2((i4s & TOP) | MAGIC)
    h2 = _h2((top & BOTTOM) | MAGIC)
    h3 = _h2((top & TOP) | MAGIC)
    k1024 = np.float16(1024.0)  # 0x6400
    sixteenth = np.float64(0.0625)  # 0x2c00
    neg64 = np.float64(-64.0)  # 0xd400
    r0 = (h0.astype(np.float64) - np.float64(k1024)).astype(np.float16)
    r1 = (h1.astype(np.float64) * sixteenth + neg64).astype(np.float16)
    r2 = (h2.astype(np.float64) - np.float64(k1024)).astype(np.float16)
    r3 = (h3.astype(np.float64) * sixteenth + neg64).astype(np.float16)
    return np.concatenate([r0, r1, r2, r3], axis=-1)


# --------------------------------------------------------------------------
# shape checks shared by the ops
# --------------------------------------------------------------------------
def _check(qweight, scales, qzeros, strict_reference: bool):
    K, NC = qweight.shape
    N = NC * PACK
    if scales.ndim != 2 or scales.shape[1] != N or scales.shape[0] == 0 or K % scales.shape[0]:
        raise ValueError("scales must be [K/G, N]")
    G = K // scales.shape[0]
    if qzeros.shape != (K // G, NC):
        raise ValueError("qzeros must be [K/G, N/8]")
    if strict_reference:
        # gemm_kernels.cu:412-419
        if N % 64:
            raise ValueError("OC is not multiple of cta_N = 64")
        if G % 32:
            raise ValueError("Group size should be a multiple of 32")
        if N % G:
            raise ValueError("OC is not multiple of Group size")
    return K, N, G


# --------------------------------------------------------------------------
# dequantisation
# --------------------------------------------------------------------------
def dequantize(qweight: np.ndarray, scales: np.ndarray, qzeros: np.ndarray) -> np.ndarray:
    """awq_dequantize: int32 [K, N/8], fp16 [K/G, N], int32 [K/G, N/8] -> fp16 [K, N].

    w[k, n] = fp16_round((q[k, n] - z[k // G, n]) * s[k // G, n]) with one
    rounding: `sub.f16x2` of two small integers is exact and
    `fma.rn.f16x2(d, s, 0)` rounds the product once
    (gemm_kernels.cu:153-160 and :313-320).
    """
    K, N, G = _check(qweight, scales, qzeros, strict_reference=False)
    q = unpack_words(qweight).astype(np.float32)  # [K, N]
    z = unpack_words(qzeros).astype(np.float32)  # [K/G, N]
    s = np.asarray(scales, dtype=np.float16).astype(np.float32)
    d = q.reshape(K // G, G, N) - z[:, None, :]  # exact, |d| <= 15
    # 5-bit integer times 11-bit significand is exact in fp32; one rounding to fp16.
    w = (d * s[:, None, :]).astype(np.float16)
    return w.reshape(K, N)


def dequantize_via_bits(qweight, scales, qzeros) -> np.ndarray:
    """Same result, but through the bit-level converter, as the kernel does it."""
    K, N, G = _check(qweight, scales, qzeros, strict_reference=False)
    q = s4_to_fp16x2_bits(qweight).reshape(K, N)
    z = s4_to_fp16x2_bits(qzeros).reshape(K // G, N)
    s = np.asarray(scales, dtype=np.float16)
    d = (q.reshape(K // G, G, N).astype(np.float32) - z[:, None, :].astype(np.float32)).astype(np.float16)
    w = (d.astype(np.float32) * s[:, None, :].astype(np.float32)).astype(np.float16)
    return w.reshape(K, N)


# --------------------------------------------------------------------------
# GEMM
# --------------------------------------------------------------------------
def gemm_faithful(x, qweight, scales, qzeros, split_k_iters: int = 8,
                  strict_reference: bool = False) -> np.ndarray:
    """awq_gemm as the reference computes it, up to fp32 re-association.

    K is cut into 32-wide chunks; chunk c belongs to split-K slice c % S
    (gemm_kernels.cu:112-115).  Each slice accumulates in fp32 and is rounded
    to fp16 (:262-270); the S slices are then summed in fp32 and rounded once
    more (`sum(0)`, :445).
    """
    x = np.asarray(x, dtype=np.float16)
    K, N, G = _check(qweight, scales, qzeros, strict_reference)
    if x.ndim != 2 or x.shape[1] != K:
        raise ValueError("x must be [M, K]")
    if K % 32:
        raise ValueError("K must be a multiple of 32")
    S = int(split_k_iters)
    w = dequantize(qweight, scales, qzeros).astype(np.float32)
    xf = x.astype(np.float32)
    M = x.shape[0]
    chunk_slice = (np.arange(K) // 32) % S
    total = np.zeros((M, N), dtype=np.float32)
    for z in range(S):
        sel = chunk_slice == z
        if not sel.any():
            continue
        part = (xf[:, sel] @ w[sel, :]).astype(np.float16)
        total += part.astype(np.float32)
    return total.astype(np.float16)


def gemm_exact(x, qweight, scales, qzeros) -> np.ndarray:
    """Same weights (fp16-rounded, as the reference), float64 accumulation, fp64 result."""
    x = np.asarray(x, dtype=np.float16)
    w = dequantize(qweight, scales, qzeros).astype(np.float64)
    return x.astype(np.float64) @ w


# --------------------------------------------------------------------------
# bf16 activations: an extension with NO reference counterpart (the reference's AWQ path is fp16 only,
# vllm/model_executor/layers/quantization/awq.py:42-43): parity unpinned by construction.  What is restated here is
# the extension's own definition: w = bf16_rne((q - z) * s) with the fp16 scale, one rounding from the exact
# product; y = sum_k w * x in float64 (the kernel accumulates in fp32 and rounds the sum to bf16 once).
# bf16 values travel as uint16 bit patterns (numpy has no bf16).
# --------------------------------------------------------------------------
def f32_to_bf16_bits(v: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even of float32 values to bf16 bit patterns (finite inputs)."""
    u = np.ascontiguousarray(v, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return r.astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def dequantize_bf16(qweight: np.ndarray, scales: np.ndarray, qzeros: np.ndarray) -> np.ndarray:
    """bf16 bit patterns [K, N] of bf16_rne((q - z) * s): the product of a 5-bit integer and an 11-bit significand
    is exact in fp32."""
    K, N, G = _check(qweight, scales, qzeros, strict_reference=False)
    q = unpack_words(qweight).astype(np.float32)
    z = unpack_words(qzeros).astype(np.float32)
    s = np.asarray(scales, dtype=np.float16).astype(np.float32)
    d = q.reshape(K // G, G, N) - z[:, None, :]
    return f32_to_bf16_bits((d * s[:, None, :]).reshape(K, N))


def gemm_bf16_exact(x_bits: np.ndarray, qweight, scales, qzeros) -> np.ndarray:
    """float64 [M, N]: bf16 activations (bit patterns) times the bf16 weights above, float64 accumulation."""
    w = bf16_bits_to_f32(dequantize_bf16(qweight, scales, qzeros)).astype(np.float64)
    return bf16_bits_to_f32(x_bits).astype(np.float64) @ w


def gemm_bf16_abs(x_bits: np.ndarray, qweight, scales, qzeros) -> np.ndarray:
    """sum_k |w * x| per output, float64: scales the bound on the kernel's fp32 accumulation error."""
    w = np.abs(bf16_bits_to_f32(dequantize_bf16(qweight, scales, qzeros)).astype(np.float64))
    return np.abs(bf16_bits_to_f32(x_bits).astype(np.float64)) @ w


# --------------------------------------------------------------------------
# synthetic inputs (seeded), shared by tests, fixtures and the benchmark
# --------------------------------------------------------------------------
def make_inputs(M: int, K: int, N: int, G: int, seed: int = 0, x_scale: float = 0.25):
    """qweight/qzeros uniform over all 32-bit patterns, scales U(1e-3, 2e-2), x N(0,1)*x_scale."""
    rng = np.random.default_rng(seed)
    qweight = rng.integers(0, 2**32, size=(K, N // PACK), dtype=np.uint64).astype(np.uint32).view(np.int32)
    qzeros = rng.integers(0, 2**32, size=(K // G, N // PACK), dtype=np.uint64).astype(np.uint32).view(np.int32)
    scales = rng.uniform(1e-3, 2e-2, size=(K // G, N)).astype(np.float16)
    x = (rng.standard_normal(size=(M, K)) * x_scale).astype(np.float16)
    return x, qweight, scales, qzeros
