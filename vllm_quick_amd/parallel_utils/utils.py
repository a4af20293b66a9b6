"""Small shard-arithmetic helpers.

Behaviour follows vllm/model_executor/parallel_utils/utils.py:12-48 of the reference.
"""
from typing import Sequence

import torch


def ensure_divisibility(numerator: int, denominator: int) -> None:
    if numerator % denominator != 0:
        raise AssertionError(f"{numerator} is not divisible by {denominator}")


def divide(numerator: int, denominator: int) -> int:
    ensure_divisibility(numerator, denominator)
    return numerator // denominator


def split_tensor_along_last_dim(tensor: torch.Tensor, num_partitions: int,
                                contiguous_split_chunks: bool = False) -> Sequence[torch.Tensor]:
    """Equal split of the last dimension; views unless `contiguous_split_chunks`."""
    width = divide(tensor.size(-1), num_partitions)
    parts = torch.split(tensor, width, dim=-1)
    if contiguous_split_chunks:
        return tuple(p.contiguous() for p in parts)
    return parts
