"""Linear layers that call a pluggable "linear method" and shard it Megatron-style.

Host-side mirror of vllm/model_executor/layers/linear.py in the reference:
same class names, constructor arguments, `forward` return convention
`(output, output_bias)`, parameter attributes (`input_dim`, `output_dim`,
`packed_dim`, `pack_factor`, `weight_loader`) and shard arithmetic, so that a
model definition written against the reference runs unchanged on top of the
W4A16 method in `layers/quantization/awq.py`.  The code is organised around one
helper (`_copy_shard`) instead of the reference's per-class loaders.

Reference lines: LinearMethodBase :20-37, UnquantizedLinearMethod :40-71,
ReplicatedLinear :74-124, ColumnParallelLinear :127-215,
MergedColumnParallelLinear :218-309, QKVParallelLinear :312-437,
RowParallelLinear :440-552.
"""
from __future__ import annotations

import logging
from abc import ABC, abstractmethod
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch.nn.parameter import Parameter

from ..parallel_utils.communication_op import (tensor_model_parallel_all_gather, tensor_model_parallel_all_reduce)
from ..parallel_utils.parallel_state import (get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size)
from ..parallel_utils.utils import divide, split_tensor_along_last_dim

logger = logging.getLogger(__name__)


def set_weight_attrs(weight: torch.Tensor, weight_attrs: Optional[Dict[str, Any]]) -> None:
    """Attach loader metadata to a parameter; refuses to overwrite (vllm/model_executor/utils.py:17-35)."""
    if not weight_attrs:
        return
    for key, value in weight_attrs.items():
        if hasattr(weight, key):
            raise AssertionError(f"Overwriting existing tensor attribute: {key}")
        setattr(weight, key, value)


class LinearMethodBase(ABC):
    """Plugin interface every (possibly quantised) linear method implements."""

    @abstractmethod
    def create_weights(self, input_size_per_partition: int, output_size_per_partition: int, input_size: int,
                       output_size: int, params_dtype: torch.dtype) -> Dict[str, Any]:
        """Allocate this rank's parameters (and any non-tensor state) for one layer."""
        raise NotImplementedError

    @abstractmethod
    def apply_weights(self, weights: Dict[str, torch.Tensor], x: torch.Tensor,
                      bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        """y = x @ W (+ bias) for the parameters `create_weights` returned."""
        raise NotImplementedError


class UnquantizedLinearMethod(LinearMethodBase):
    """Dense fp16/bf16/fp32 weights; the default when a layer gets no linear method."""

    def __init__(self, separate_bias_add: bool = False):
        self.separate_bias_add = separate_bias_add

    def create_weights(self, input_size_per_partition, output_size_per_partition, input_size, output_size,
                       params_dtype):
        weight = Parameter(torch.empty(output_size_per_partition, input_size_per_partition, dtype=params_dtype),
                           requires_grad=False)
        set_weight_attrs(weight, {"input_dim": 1, "output_dim": 0})
        return {"weight": weight}

    def apply_weights(self, weights, x, bias=None):
        weight = weights["weight"]
        if self.separate_bias_add:
            y = F.linear(x, weight)
            return y + bias if bias is not None else y
        return F.linear(x, weight, bias)


# --------------------------------------------------------------------------
# shared plumbing
# --------------------------------------------------------------------------
def _packed_units(param: torch.Tensor, dim: Optional[int], offset: int, size: int) -> Tuple[int, int]:
    """Offsets along a dimension that the parameter stores `pack_factor` elements per word."""
    if dim is not None and getattr(param, "packed_dim", None) == dim:
        pf = param.pack_factor
        return offset // pf, size // pf
    return offset, size


def _copy_shard(dst: torch.Tensor, src: torch.Tensor, owner: Optional[torch.Tensor] = None) -> None:
    """Copy a checkpoint shard into (a slice of) a parameter's storage.

    Loaders write through `param.data`, as the reference's do; such writes leave the parameter's
    in-place-write counter alone, and linear methods that keep a derived copy of the weights (the
    W4A16 method's MFMA-ordered layout) watch exactly that counter.  So the owner is marked as
    written by hand.
    """
    if dst.shape != src.shape:
        raise AssertionError(f"shard shape mismatch: parameter {tuple(dst.shape)} vs checkpoint {tuple(src.shape)}")
    dst.copy_(src)
    if owner is not None:
        mark_written(owner)


# Counts the writes the weight loaders have made, process-wide: modules that keep a derived copy of many
# tensors (the expert block) look at this one integer before they look at every tensor.
_WRITE_EPOCH = [0]


def write_epoch() -> int:
    return _WRITE_EPOCH[0]


def mark_written(param: torch.Tensor) -> None:
    """Bump a tensor's in-place-write counter by one after a write that bypassed it (`param.data...`)."""
    _WRITE_EPOCH[0] += 1
    bump = getattr(torch._C, "_increment_version", None)
    if bump is None:
        return
    try:
        bump([param])  # current torch: an iterable of tensors (a bare tensor would be walked row by row)
    except TypeError:
        try:
            bump(param)  # older torch: one tensor
        except RuntimeError:
            pass
    except RuntimeError:  # inference tensors keep no counter
        pass


class _LinearBase(torch.nn.Module):
    """Parameter creation shared by all layer flavours."""

    def _setup(self, in_per_rank: int, out_per_rank: int, input_size: int, output_size: int,
               params_dtype: Optional[torch.dtype], linear_method: Optional[LinearMethodBase],
               attach_loader: bool) -> None:
        self.params_dtype = params_dtype if params_dtype is not None else torch.get_default_dtype()
        self.linear_method = linear_method if linear_method is not None else UnquantizedLinearMethod()
        self.linear_weights = self.linear_method.create_weights(in_per_rank, out_per_rank, input_size, output_size,
                                                                self.params_dtype)
        for name, value in self.linear_weights.items():
            if isinstance(value, torch.Tensor):
                self.register_parameter(name, value)
                if attach_loader:
                    set_weight_attrs(value, {"weight_loader": self.weight_loader})

    def _make_bias(self, enabled: bool, width: int, attach_loader: bool) -> None:
        if not enabled:
            self.register_parameter("bias", None)
            return
        self.bias = Parameter(torch.empty(width, dtype=self.params_dtype))
        attrs: Dict[str, Any] = {"output_dim": 0}
        if attach_loader:
            attrs["weight_loader"] = self.weight_loader
        set_weight_attrs(self.bias, attrs)


class ReplicatedLinear(_LinearBase):
    """Y = XA + b with the whole of A on every rank."""

    def __init__(self, input_size: int, output_size: int, bias: bool = True, skip_bias_add: bool = False,
                 params_dtype: Optional[torch.dtype] = None, linear_method: Optional[LinearMethodBase] = None):
        super().__init__()
        self.input_size = input_size
        self.output_size = output_size
        self.skip_bias_add = skip_bias_add
        self._setup(input_size, output_size, input_size, output_size, params_dtype, linear_method,
                    attach_loader=False)
        self._make_bias(bias, output_size, attach_loader=False)

    def forward(self, x: torch.Tensor):
        fused_bias = None if self.skip_bias_add else self.bias
        out = self.linear_method.apply_weights(self.linear_weights, x, fused_bias)
        return out, (self.bias if self.skip_bias_add else None)


class ColumnParallelLinear(_LinearBase):
    """A = [A_1 .. A_p] split along the output dimension; rank i computes X A_i."""

    def __init__(self, input_size: int, output_size: int, bias: bool = True, gather_output: bool = False,
                 skip_bias_add: bool = False, params_dtype: Optional[torch.dtype] = None,
                 linear_method: Optional[LinearMethodBase] = None):
        super().__init__()
        self.input_size = input_size
        self.output_size = output_size
        self.gather_output = gather_output
        self.skip_bias_add = skip_bias_add
        tp = get_tensor_model_parallel_world_size()
        self.output_size_per_partition = divide(output_size, tp)
        self._setup(input_size, self.output_size_per_partition, input_size, output_size, params_dtype, linear_method,
                    attach_loader=True)
        self._make_bias(bias, self.output_size_per_partition, attach_loader=True)

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor) -> None:
        out_dim = getattr(param, "output_dim", None)
        if out_dim is not None:
            width = param.data.shape[out_dim]
            loaded_weight = loaded_weight.narrow(out_dim, get_tensor_model_parallel_rank() * width, width)
        _copy_shard(param.data, loaded_weight, param)

    def forward(self, input_: torch.Tensor):
        fused_bias = None if self.skip_bias_add else self.bias
        local = self.linear_method.apply_weights(self.linear_weights, input_, fused_bias)
        out = tensor_model_parallel_all_gather(local) if self.gather_output else local
        return out, (self.bias if self.skip_bias_add else None)


class MergedColumnParallelLinear(ColumnParallelLinear):
    """Several column-parallel matrices fused along the output dimension (e.g. gate_proj + up_proj).

    Each sub-matrix is sharded on its own, so rank i holds [A1_i | A2_i | ...].
    """

    def __init__(self, input_size: int, output_sizes: List[int], bias: bool = True, gather_output: bool = False,
                 skip_bias_add: bool = False, params_dtype: Optional[torch.dtype] = None,
                 linear_method: Optional[LinearMethodBase] = None):
        self.output_sizes = list(output_sizes)
        tp = get_tensor_model_parallel_world_size()
        if any(size % tp for size in self.output_sizes):
            raise AssertionError(f"output sizes {self.output_sizes} are not all divisible by TP size {tp}")
        super().__init__(input_size, sum(self.output_sizes), bias, gather_output, skip_bias_add, params_dtype,
                         linear_method)

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor,
                      loaded_shard_id: Optional[int] = None) -> None:
        out_dim = getattr(param, "output_dim", None)
        if loaded_shard_id is None:
            # The checkpoint stores the fused matrix: peel the sub-matrices off and load each.
            if out_dim is None:
                _copy_shard(param.data, loaded_weight, param)
                return
            start = 0
            for idx, size in enumerate(self.output_sizes):
                off, width = _packed_units(param, out_dim, start, size)
                self.weight_loader(param, loaded_weight.narrow(out_dim, off, width), idx)
                start += size
            return
        if not 0 <= loaded_shard_id < len(self.output_sizes):
            raise AssertionError(f"shard id {loaded_shard_id} out of range")
        dst = param.data
        if out_dim is not None:
            tp, rank = get_tensor_model_parallel_world_size(), get_tensor_model_parallel_rank()
            off, width = _packed_units(param, out_dim, sum(self.output_sizes[:loaded_shard_id]) // tp,
                                       self.output_sizes[loaded_shard_id] // tp)
            dst = dst.narrow(out_dim, off, width)
            loaded_weight = loaded_weight.narrow(out_dim, rank * width, width)
        elif not getattr(param, "ignore_warning", False):
            logger.warning("Loading a weight without `output_dim` attribute in MergedColumnParallelLinear, "
                           "assume the weight is the same for all partitions.")
        _copy_shard(dst, loaded_weight, param)


class QKVParallelLinear(ColumnParallelLinear):
    """Fused q/k/v projection, parallel over heads; kv heads are replicated when TP exceeds their count."""

    def __init__(self, hidden_size: int, head_size: int, total_num_heads: int,
                 total_num_kv_heads: Optional[int] = None, bias: bool = True, skip_bias_add: bool = False,
                 params_dtype: Optional[torch.dtype] = None, linear_method: Optional[LinearMethodBase] = None):
        self.hidden_size = hidden_size
        self.head_size = head_size
        self.total_num_heads = total_num_heads
        self.total_num_kv_heads = total_num_heads if total_num_kv_heads is None else total_num_kv_heads
        tp = get_tensor_model_parallel_world_size()
        self.num_heads = divide(self.total_num_heads, tp)
        if tp >= self.total_num_kv_heads:
            self.num_kv_heads = 1
            self.num_kv_head_replicas = divide(tp, self.total_num_kv_heads)
        else:
            self.num_kv_heads = divide(self.total_num_kv_heads, tp)
            self.num_kv_head_replicas = 1
        output_size = (self.num_heads + 2 * self.num_kv_heads) * tp * self.head_size
        super().__init__(hidden_size, output_size, bias, False, skip_bias_add, params_dtype, linear_method)

    def _local_span(self, which: str) -> Tuple[int, int]:
        """(offset, width) of q, k or v inside this rank's fused output, in elements."""
        q, kv = self.num_heads * self.head_size, self.num_kv_heads * self.head_size
        return {"q": (0, q), "k": (q, kv), "v": (q + kv, kv)}[which]

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor,
                      loaded_shard_id: Optional[str] = None) -> None:
        out_dim = getattr(param, "output_dim", None)
        if loaded_shard_id is None:
            if out_dim is None:
                _copy_shard(param.data, loaded_weight, param)
                return
            q_all = self.total_num_heads * self.head_size
            kv_all = self.total_num_kv_heads * self.head_size
            for which, start, size in (("q", 0, q_all), ("k", q_all, kv_all), ("v", q_all + kv_all, kv_all)):
                off, width = _packed_units(param, out_dim, start, size)
                self.weight_loader(param, loaded_weight.narrow(out_dim, off, width), which)
            return
        if loaded_shard_id not in ("q", "k", "v"):
            raise AssertionError(f"unknown shard id {loaded_shard_id!r}")
        dst = param.data
        if out_dim is not None:
            off, width = _packed_units(param, out_dim, *self._local_span(loaded_shard_id))
            dst = dst.narrow(out_dim, off, width)
            rank = get_tensor_model_parallel_rank()
            src_block = rank if loaded_shard_id == "q" else rank // self.num_kv_head_replicas
            loaded_weight = loaded_weight.narrow(out_dim, src_block * width, width)
        elif not getattr(param, "ignore_warning", False):
            logger.warning("Loading a weight without `output_dim` attribute in QKVParallelLinear, "
                           "assume the weight is the same for all partitions.")
        _copy_shard(dst, loaded_weight, param)


class RowParallelLinear(_LinearBase):
    """A split along the input dimension; rank i computes X_i A_i and the partials are all-reduced."""

    def __init__(self, input_size: int, output_size: int, bias: bool = True, input_is_parallel: bool = True,
                 skip_bias_add: bool = False, params_dtype: Optional[torch.dtype] = None,
                 reduce_results: bool = True, linear_method: Optional[LinearMethodBase] = None):
        super().__init__()
        self.input_size = input_size
        self.output_size = output_size
        self.input_is_parallel = input_is_parallel
        self.reduce_results = reduce_results
        self.skip_bias_add = skip_bias_add
        self.tp_size = get_tensor_model_parallel_world_size()
        self.input_size_per_partition = divide(input_size, self.tp_size)
        self._setup(self.input_size_per_partition, output_size, input_size, output_size, params_dtype, linear_method,
                    attach_loader=True)
        if not reduce_results and (bias and not skip_bias_add):
            raise ValueError("When not reduce the results, adding bias to the results can lead to incorrect results")
        self._make_bias(bias, output_size, attach_loader=True)

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor) -> None:
        in_dim = getattr(param, "input_dim", None)
        if in_dim is not None:
            depth = param.data.shape[in_dim]
            loaded_weight = loaded_weight.narrow(in_dim, get_tensor_model_parallel_rank() * depth, depth)
        _copy_shard(param.data, loaded_weight, param)

    def forward(self, input_: torch.Tensor):
        if self.input_is_parallel:
            local_in = input_
        else:
            pieces = split_tensor_along_last_dim(input_, num_partitions=self.tp_size)
            local_in = pieces[get_tensor_model_parallel_rank()].contiguous()
        # The bias is not sharded, so it is added once, after the reduction.
        partial = self.linear_method.apply_weights(self.linear_weights, local_in)
        out = tensor_model_parallel_all_reduce(partial) if (self.reduce_results and self.tp_size > 1) else partial
        if self.skip_bias_add:
            return out, self.bias
        return (out + self.bias if self.bias is not None else out), None


def row_parallel_add_(layer: "RowParallelLinear", x: torch.Tensor, residual: torch.Tensor) -> torch.Tensor:
    """residual += layer(x)[0], in place; returns `residual`.

    What a decoder layer does with its o_proj and down_proj outputs (llama.py:197-223).  On one rank,
    with a linear method that offers it (AWQ), the addition runs in the GEMM's epilogue; with tensor
    parallelism the partial products must be summed over the ranks first, so the two-step route is taken.
    """
    fused = getattr(layer.linear_method, "apply_weights_add_", None)
    if fused is None or layer.tp_size > 1 or layer.skip_bias_add:
        y, _ = layer(x)
        return residual.add_(y)
    return fused(layer.linear_weights, x, residual, layer.bias)
