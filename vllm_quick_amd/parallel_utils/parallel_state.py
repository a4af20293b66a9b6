"""Tensor-model-parallel group bookkeeping.

One process per GPU; the TP group is a `torch.distributed` group whose backend
is "nccl" on GPUs (RCCL on ROCm, riding the xGMI mesh) or "gloo" in CPU tests.
The reference builds Megatron-style TP and (unused) PP groups
(vllm/model_executor/parallel_utils/parallel_state.py:36-129); only the TP
group matters for the W4A16 linear layers, so that is what is kept, with the
same accessor names.
"""
from __future__ import annotations

import contextlib
from typing import Optional

import torch.distributed as dist

_TP_GROUP: Optional[dist.ProcessGroup] = None
_TP_RANKS: Optional[list] = None


def initialize_model_parallel(tensor_model_parallel_size: int = 1) -> None:
    """Split the world into consecutive-rank TP groups of the given size (parallel_state.py:69-74)."""
    global _TP_GROUP, _TP_RANKS
    if not dist.is_initialized():
        if tensor_model_parallel_size != 1:
            raise RuntimeError("torch.distributed must be initialised for tensor parallelism > 1")
        return
    world = dist.get_world_size()
    if world % tensor_model_parallel_size != 0:
        raise RuntimeError(f"world size {world} is not divisible by TP size {tensor_model_parallel_size}")
    if _TP_GROUP is not None:
        raise AssertionError("tensor model parallel group is already initialized")
    rank = dist.get_rank()
    for g in range(world // tensor_model_parallel_size):
        ranks = list(range(g * tensor_model_parallel_size, (g + 1) * tensor_model_parallel_size))
        group = dist.new_group(ranks)  # every rank must create every group
        if rank in ranks:
            _TP_GROUP, _TP_RANKS = group, ranks


def model_parallel_is_initialized() -> bool:
    return _TP_GROUP is not None


def get_tensor_model_parallel_group() -> Optional[dist.ProcessGroup]:
    return _TP_GROUP


def get_tensor_model_parallel_world_size() -> int:
    if _REHEARSED is not None:
        return _REHEARSED[0]
    if _TP_GROUP is None:
        return 1
    return dist.get_world_size(group=_TP_GROUP)


def get_tensor_model_parallel_rank() -> int:
    if _REHEARSED is not None:
        return _REHEARSED[1]
    if _TP_GROUP is None:
        return 0
    return dist.get_rank(group=_TP_GROUP)


# One rank's share of a tensor-parallel model on a single process, for measuring the compute half of a step
# without the other ranks: layers built inside the block shard as rank `rank` of `world` would, and
# `tensor_model_parallel_all_reduce` becomes a device copy of the same size (no sum: results are one rank's
# partial products, good for timing only).
_REHEARSED: Optional[tuple] = None


@contextlib.contextmanager
def rehearse_one_rank(world: int, rank: int = 0):
    global _REHEARSED
    if _TP_GROUP is not None:
        raise AssertionError("a tensor model parallel group is initialized: nothing to rehearse")
    if not 0 <= rank < world:
        raise ValueError("rank must be in [0, world)")
    prev, _REHEARSED = _REHEARSED, (world, rank)
    try:
        yield
    finally:
        _REHEARSED = prev


def rehearsing_one_rank() -> bool:
    return _REHEARSED is not None


def get_tensor_model_parallel_src_rank() -> int:
    """Global rank of the first member of this process's TP group."""
    if _TP_RANKS is None:
        return 0
    return _TP_RANKS[0]


def destroy_model_parallel() -> None:
    global _TP_GROUP, _TP_RANKS
    if _TP_GROUP is not None and dist.is_initialized():
        dist.destroy_process_group(_TP_GROUP)
    _TP_GROUP, _TP_RANKS = None, None
    from . import rccl_utils
    rccl_utils.destroy_process_group()  # the direct communicator, if any (parallel_state.py:213-214)


# Whether all-reduces go to RCCL directly instead of through torch.distributed: needed while a
# HIP graph is captured, where ProcessGroupNCCL's watchdog aborts the process
# (parallel_state.py:217-245 does the same with CuPy's NCCL binding, for the same reason).
_ENABLE_DIRECT_RCCL_FOR_ALL_REDUCE = False


@contextlib.contextmanager
def with_rccl_for_all_reduce():
    """Use the direct RCCL communicator (rccl_utils) for all-reduce inside this block; each call
    goes to the stream that is current when it is made (the capturing stream, under capture)."""
    global _ENABLE_DIRECT_RCCL_FOR_ALL_REDUCE
    if get_tensor_model_parallel_world_size() == 1:
        yield  # nothing to reduce; the communicator is not even created
        return
    old = _ENABLE_DIRECT_RCCL_FOR_ALL_REDUCE
    _ENABLE_DIRECT_RCCL_FOR_ALL_REDUCE = True
    try:
        yield
    finally:
        _ENABLE_DIRECT_RCCL_FOR_ALL_REDUCE = old


def is_rccl_enabled_for_all_reduce() -> bool:
    return _ENABLE_DIRECT_RCCL_FOR_ALL_REDUCE


# the reference's names, so that its call sites (vllm/worker/model_runner.py capture loop) read the same
with_cupy_nccl_for_all_reduce = with_rccl_for_all_reduce
is_cupy_nccl_enabled_for_all_reduce = is_rccl_enabled_for_all_reduce
