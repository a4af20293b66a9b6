"""Tensor-model-parallel group bookkeeping.

One process per GPU; the TP group is a `torch.distributed` group whose backend
is "nccl" on GPUs (RCCL on ROCm, riding the xGMI mesh) or "gloo" in CPU tests.
The reference builds Megatron-style TP and (unused) PP groups
(vllm/model_executor/parallel_utils/parallel_state.py:36-129); only the TP
group matters for the W4A16 linear layers, so that is what is kept, with the
same accessor names.
"""
from __future__ import annotations

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
    if _TP_GROUP is None:
        return 1
    return dist.get_world_size(group=_TP_GROUP)


def get_tensor_model_parallel_rank() -> int:
    if _TP_GROUP is None:
        return 0
    return dist.get_rank(group=_TP_GROUP)


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
