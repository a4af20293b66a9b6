"""Collectives of the tensor-parallel linear layers.

Reference: vllm/model_executor/parallel_utils/communication_op.py:17-69.  As there, an all-reduce
first tries the one-kernel peer-to-peer path (`custom_all_reduce`, active once `init_custom_ar()`
has run for the group and for inputs it accepts) and otherwise goes to torch.distributed: RCCL
over xGMI on GPUs, gloo in CPU tests.  Inside `with_rccl_for_all_reduce()` (graph capture) RCCL is
called directly (`rccl_utils`), the counterpart of the reference's CuPy-NCCL leg.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .custom_all_reduce import custom_all_reduce
from . import rccl_utils
from .parallel_state import (get_tensor_model_parallel_group, get_tensor_model_parallel_world_size,
                             is_rccl_enabled_for_all_reduce, rehearsing_one_rank)


def tensor_model_parallel_all_reduce(input_: torch.Tensor) -> torch.Tensor:
    """Sum over the TP group.  May or may not work in place: always assume the input is modified
    and use the return value (communication_op.py:17-41)."""
    if get_tensor_model_parallel_world_size() == 1:
        return input_
    if rehearsing_one_rank():
        return input_.clone()  # one rank alone (parallel_state.rehearse_one_rank): a device copy of the same size
    out = custom_all_reduce(input_)
    if out is not None:
        return out
    if is_rccl_enabled_for_all_reduce():
        rccl_utils.all_reduce(input_)  # capturable: no helper thread watches it
    else:
        dist.all_reduce(input_, group=get_tensor_model_parallel_group())
    return input_


def tensor_model_parallel_all_gather(input_: torch.Tensor, dim: int = -1) -> torch.Tensor:
    """Concatenate every rank's tensor along `dim` (rank order)."""
    world = get_tensor_model_parallel_world_size()
    if world == 1:
        return input_
    if not -input_.dim() <= dim < input_.dim():
        raise AssertionError(f"Invalid dim ({dim}) for input tensor with shape {input_.size()}")
    dim %= input_.dim()
    shape = tuple(input_.shape)
    if not shape:
        raise AssertionError("cannot all-gather a 0-d tensor")
    # Concatenated-along-dim-0 form: accepted by both RCCL and gloo.
    flat = torch.empty((world * shape[0], ) + shape[1:], dtype=input_.dtype, device=input_.device)
    dist.all_gather_into_tensor(flat, input_.contiguous(), group=get_tensor_model_parallel_group())
    gathered = flat.view((world, ) + shape)
    # [world, ..., d, ...] -> [..., world, d, ...] -> [..., world * d, ...]
    gathered = gathered.movedim(0, dim)
    return gathered.reshape(shape[:dim] + (world * shape[dim], ) + shape[dim + 1:])
