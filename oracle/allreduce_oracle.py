"""CPU oracle for the reference's custom all-reduce arithmetic (numpy).

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package imports this module.

PARITY PINNED ONLY FOR EXACT SUMS: the reference's one test of this path
(tests/distributed/test_custom_all_reduce.py:30-53, 64-73) uses small integers and ones, whose sum
is exact in every order and equals NCCL's; `test_allreduce_oracle_cpu.py` checks this file against
that property.  For inputs whose fp32 partial sums round, the ORDER of the additions matters; it is
restated here from the reference's sources and no vector of the reference pins it.

What is restated (paths relative to the reference checkout):

* element-wise fp32 accumulation, one rounding at the end     csrc/custom_all_reduce.cuh:204-212
* one-stage order: ranks 0, 1, ..                              csrc/custom_all_reduce.cuh:221-230
* two-stage: parts of floor(units / world) 16-byte units, the last part takes the remainder;
  part p summed in the order p, p+1, .. (mod world)            csrc/custom_all_reduce.cuh:248-264,270-282
* which of the two runs                                        csrc/custom_all_reduce.cuh:520-528
* which inputs take the path at all                            csrc/custom_all_reduce.cu:58-68
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

UNIT_BYTES = 16
DTYPES = {"f32": 4, "f16": 2, "bf16": 2}


def bf16_to_f32(bits: np.ndarray) -> np.ndarray:
    return (bits.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16(x: np.ndarray) -> np.ndarray:
    """Round to nearest even; NaN stays NaN."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    nan = (u & 0x7fffffff) > 0x7f800000
    r = ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    return np.where(nan, ((u >> 16) | 0x40).astype(np.uint16), r)


def _up(a: np.ndarray, dtype: str) -> np.ndarray:
    if dtype == "f32":
        return a.astype(np.float32)
    if dtype == "f16":
        return a.view(np.float16).astype(np.float32)
    return bf16_to_f32(a.view(np.uint16))


def _down(acc: np.ndarray, dtype: str) -> np.ndarray:
    if dtype == "f32":
        return acc.astype(np.float32)
    if dtype == "f16":
        with np.errstate(over="ignore"):
            return acc.astype(np.float16)
    return f32_to_bf16(acc)


def should_custom_ar(nbytes: int, max_size: int, world: int, full_mesh: bool) -> bool:
    if nbytes % UNIT_BYTES != 0:
        return False
    if world == 2 or full_mesh:
        return nbytes <= max_size
    return world <= 4 and nbytes <= 512 * 1024


def uses_one_stage(world: int, nbytes: int) -> bool:
    return world == 2 or (world <= 4 and nbytes < 512 * 1024) or (world <= 8 and nbytes < 256 * 1024)


def _sum_in_order(parts: Sequence[np.ndarray], order: Sequence[int], dtype: str) -> np.ndarray:
    acc = _up(parts[order[0]], dtype)
    for r in order[1:]:
        acc = (acc + _up(parts[r], dtype)).astype(np.float32)
    return _down(acc, dtype)


def all_reduce(inputs: List[np.ndarray], dtype: str) -> np.ndarray:
    """inputs[r] = rank r's flat vector (float32 / float16 arrays, or uint16 bit patterns for bf16).
    Returns what every rank holds afterwards, in the same representation."""
    world = len(inputs)
    esize = DTYPES[dtype]
    n = inputs[0].size
    nbytes = n * esize
    assert nbytes % UNIT_BYTES == 0 and all(a.size == n for a in inputs)
    flat = [np.ascontiguousarray(a).reshape(-1) for a in inputs]
    if uses_one_stage(world, nbytes):
        return _sum_in_order(flat, list(range(world)), dtype)
    per_unit = UNIT_BYTES // esize
    units = nbytes // UNIT_BYTES
    part = units // world
    out = np.empty_like(flat[0])
    for p in range(world):
        lo = p * part * per_unit
        hi = n if p == world - 1 else (p + 1) * part * per_unit
        order = [(p + k) % world for k in range(world)]
        out[lo:hi] = _sum_in_order([a[lo:hi] for a in flat], order, dtype)
    return out
