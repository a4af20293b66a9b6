"""Gated-MLP activation next to the W4A16 GEMM.

`SiluAndMul` mirrors the reference layer of that name (vllm/model_executor/layers/activation.py:13-35:
x -> silu(x[..., :d]) * x[..., d:], d = x.shape[-1] // 2).  `gate_up_silu_and_mul` applies a merged
gate_up projection and that activation in one step; with an AWQ linear method the activation runs in the
GEMM's epilogue and the [tokens, 2 d] intermediate never reaches memory.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .linear import ColumnParallelLinear


class SiluAndMul(torch.nn.Module):

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        d = x.shape[-1] // 2
        return F.silu(x[..., :d]) * x[..., d:]


def gate_up_silu_and_mul(gate_up_proj: ColumnParallelLinear, x: torch.Tensor) -> torch.Tensor:
    """act_fn(gate_up_proj(x)[0]) of llama.py:77-78 for a `MergedColumnParallelLinear([d, d])`.

    Each rank holds [gate_i | up_i] (the merged layer shards every sub-matrix on its own), which is the
    form the fused epilogue expects.  Layers that gather their output, skip the bias add or use a linear
    method without the fused call take the two-step route.
    """
    method = gate_up_proj.linear_method
    fused = getattr(method, "apply_weights_silu_and_mul", None)
    if fused is None or gate_up_proj.gather_output or (gate_up_proj.skip_bias_add and gate_up_proj.bias is not None):
        y, _ = gate_up_proj(x)
        return SiluAndMul()(y)
    return fused(gate_up_proj.linear_weights, x, gate_up_proj.bias)
