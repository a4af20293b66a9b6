"""Peer-to-peer all-reduce without a GPU: the oracle against the property the reference's own test
uses (integer inputs sum exactly, tests/distributed/test_custom_all_reduce.py:30-53,64-73), the
C ABI's exports and host-only entry points, and the host mirror's decision logic."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HEADER = os.path.join(ROOT, "include", "p2p_allreduce.h")

from oracle import allreduce_oracle as aro  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    from vllm_quick_amd import _p2p_lib
    return _p2p_lib


def to_dtype(ints, dtype):
    if dtype == "f32":
        return ints.astype(np.float32)
    if dtype == "f16":
        return ints.astype(np.float16)
    return aro.f32_to_bf16(ints.astype(np.float32))


def as_f32(a, dtype):
    return aro.bf16_to_f32(a) if dtype == "bf16" else a.astype(np.float32)


@pytest.mark.parametrize("world", [2, 4, 6, 8])
@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("n", [8, 1032, 300 * 1024])
def test_oracle_sums_small_integers_exactly(world, dtype, n):
    rng = np.random.default_rng(world * 1000 + n)
    ints = [rng.integers(1, 16, n) for _ in range(world)]  # the reference test's value range
    got = aro.all_reduce([to_dtype(v, dtype) for v in ints], dtype)
    assert np.array_equal(as_f32(got, dtype), np.sum(ints, axis=0).astype(np.float32))


def test_oracle_orders_differ_only_by_rounding():
    """Non-integer data: one-stage and two-stage orders agree to fp32 rounding, each part of the
    two-stage result equals the sum taken in that part's own order."""
    rng = np.random.default_rng(5)
    world, n = 4, 512 * 1024 // 2  # 512 KiB of fp16: two-stage
    xs = [(rng.standard_normal(n) * 3).astype(np.float16) for _ in range(world)]
    assert not aro.uses_one_stage(world, n * 2)
    got = aro.all_reduce(xs, "f16")
    part = (n * 2 // 16 // world) * 8
    for p in range(world):
        lo, hi = p * part, (n if p == world - 1 else (p + 1) * part)
        acc = xs[p][lo:hi].astype(np.float32)
        for k in range(1, world):
            acc = acc + xs[(p + k) % world][lo:hi].astype(np.float32)
        assert np.array_equal(got[lo:hi].view(np.uint16), acc.astype(np.float16).view(np.uint16))
    exact = np.sum([x.astype(np.float64) for x in xs], axis=0)
    assert np.allclose(got.astype(np.float64), exact, rtol=2e-3, atol=2e-3)


def test_bf16_rounding_is_nearest_even():
    x = np.array([1.0, 1.00390625, 1.01171875, -2.5, 3.0e38, np.inf], dtype=np.float32)
    got = aro.bf16_to_f32(aro.f32_to_bf16(x))
    want = torch.from_numpy(x).to(torch.bfloat16).float().numpy()
    assert np.array_equal(got, want)
    assert np.isnan(aro.bf16_to_f32(aro.f32_to_bf16(np.array([np.nan], dtype=np.float32))))[0]


def test_algorithm_choice_and_admission_follow_the_reference():
    K = 1024
    assert aro.uses_one_stage(2, 8192 * K)
    assert aro.uses_one_stage(4, 512 * K - 16) and not aro.uses_one_stage(4, 512 * K)
    assert aro.uses_one_stage(8, 256 * K - 16) and not aro.uses_one_stage(8, 256 * K)
    assert aro.uses_one_stage(6, 256 * K - 16) and not aro.uses_one_stage(6, 256 * K)
    assert aro.should_custom_ar(8192 * K, 8192 * K, 8, True)
    assert not aro.should_custom_ar(8192 * K + 16, 8192 * K, 8, True)
    assert not aro.should_custom_ar(1000, 8192 * K, 2, True)  # not a multiple of 16
    assert aro.should_custom_ar(512 * K, 8192 * K, 4, False) and not aro.should_custom_ar(512 * K + 16, 8192 * K, 4,
                                                                                         False)
    assert not aro.should_custom_ar(1024, 8192 * K, 8, False)


# ------------------------------------------------------------------ C ABI, host-only entry points
def declared_symbols():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(p2p_ar_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported_and_bound(lib):
    names = declared_symbols()
    assert len(names) >= 12
    raw = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/p2p_allreduce.h but not exported"
    assert sorted(lib.SYMBOLS) == names


def test_header_cites_reference():
    text = open(HEADER).read()
    for cite in ("custom_all_reduce.cu:58-68", "custom_all_reduce.cu:12-34", "custom_all_reduce.cu:101-122",
                 "custom_all_reduce.cuh:520-528", "pybind.cpp:99-113"):
        assert cite in text


def test_should_matches_oracle(lib):
    K = 1024
    for world in (2, 4, 6, 8):
        for mesh in (0, 1):
            for nbytes in (16, 1000, 4096, 256 * K, 512 * K, 512 * K + 16, 8192 * K, 8192 * K + 16):
                want = aro.should_custom_ar(nbytes, 8192 * K, world, bool(mesh))
                assert bool(lib.lib.p2p_ar_should(nbytes, 8192 * K, world, mesh)) == want, (world, mesh, nbytes)


def test_host_only_entry_points(lib):
    L = lib.lib
    assert L.p2p_ar_abi_version() == lib.ABI_VERSION
    assert L.p2p_ar_strerror(0) == b"ok"
    assert b"(2,4,6,8)" in L.p2p_ar_strerror(-1)
    assert b"float32, float16 and bfloat16" in L.p2p_ar_strerror(-5)
    assert L.p2p_ar_mailbox_bytes(3, 1 << 20) == 0  # odd world sizes are not supported
    flags = 2 * 2 * 8 * 64 * 4
    # world 2: both halves hold two whole vectors; world 8: two stages of (max + slack) each
    assert L.p2p_ar_mailbox_bytes(2, 8 << 20) == flags + 2 * 2 * ((8 << 20) + 4096)
    assert L.p2p_ar_mailbox_bytes(8, 8 << 20) == flags + 2 * 2 * ((8 << 20) + 4096)
    # a small maximum with many ranks: the one-stage slots (world vectors) set the size
    assert L.p2p_ar_mailbox_bytes(8, 64 * 1024) == flags + 2 * 8 * 64 * 1024
    ctx = ctypes.c_void_p()
    assert L.p2p_ar_create(0, 3, 1 << 20, ctypes.byref(ctx)) == -1
    assert L.p2p_ar_create(2, 2, 1 << 20, ctypes.byref(ctx)) == -1
    assert L.p2p_ar_create(0, 2, 1000, ctypes.byref(ctx)) == -3
    assert L.p2p_ar_create(0, 2, 1 << 20, None) == -2
    assert L.p2p_ar_all_reduce(None, None, None, 8, 1, None) == -2
    with pytest.raises(ValueError, match="2,4,6,8"):
        lib.check(-1, "x")
    with pytest.raises(RuntimeError):
        lib.check(-6, "x")


# ------------------------------------------------------------------ host mirror
def test_mirror_without_a_group_is_inert():
    from vllm_quick_amd.parallel_utils import communication_op as comm
    from vllm_quick_amd.parallel_utils import custom_all_reduce as car
    assert car.get_handle() is None and not car.is_capturing()
    car.init_custom_ar()  # world size 1: nothing to set up (custom_all_reduce.py:32-34)
    assert car.get_handle() is None
    t = torch.ones(8)
    assert car.custom_all_reduce(t) is None
    with car.capture():
        assert not car.is_capturing()  # capturing only counts with a handle (custom_all_reduce.py:62-63)
    assert comm.tensor_model_parallel_all_reduce(t) is t


def test_weak_contiguity_examples_of_the_reference():
    """The six cases listed at csrc/custom_all_reduce.cu:42-50."""
    from vllm_quick_amd.parallel_utils.custom_all_reduce import _is_weak_contiguous as ok
    A = torch.zeros(3, 3, 3)
    assert ok(A) and ok(A[1:]) and ok(A.permute(2, 0, 1)) and ok(A[1:].permute(2, 0, 1))
    assert not ok(A[None].expand(2, -1, -1, -1)) and not ok(A[:, 1:, 1:])


def test_direct_rccl_context_is_a_no_op_for_one_rank():
    """parallel_state.py:226-230: nothing is switched (or created) when there is nothing to reduce."""
    from vllm_quick_amd.parallel_utils import parallel_state as ps
    from vllm_quick_amd.parallel_utils import rccl_utils
    with ps.with_rccl_for_all_reduce():
        assert not ps.is_rccl_enabled_for_all_reduce()
    assert not rccl_utils.is_initialized() and rccl_utils.get_world_size() == 0
    rccl_utils.destroy_process_group()  # harmless without a communicator
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclCommDestroy", "ncclGetErrorString"):
        assert hasattr(rccl_utils._library(), name)
