"""Quantised mixture-of-experts block on batched W4A16 launches.

Host-side mirror of `MixtralMLP` / `MixtralMoE` in the reference
(vllm/model_executor/models/mixtral_quant.py:57-93, 96-160): same constructor arguments,
the same parameter tree (`experts.<i>.w1|w2|w3.{qweight,qzeros,scales}`, `gate.weight`), the
same routing arithmetic, experts split over the tensor-parallel ranks the same way with one
all-reduce at the end.  What differs is how the local experts run.  The reference calls every
local expert on all tokens, one AWQ GEMM launch at a time (3 launches per expert, :145-157);
here the local experts' weights are kept back to back in the private layout and the block
takes three launches in total:

    [w1_e | w3_e], e = 0 .. E-1   x  tokens           E problems, shared activations, with
                                                      silu(w1 out) * w3 out fused into the epilogue
    [w2_0 .. w2_{E-1}]            x  per-expert act   E problems

followed by the routing-weighted sum over experts.  (`fuse_act = False` runs 2E separate gate/up
problems and the activation as an elementwise step instead.)

With few tokens most experts are selected by no token at all; the reference still runs them and
multiplies their output by a routing weight of zero (:150-155).  Up to `sparse_max_tokens` (4) tokens
the launches here carry a per-expert switch computed on the device from the routing weights (no
host synchronisation, so the block stays capturable in a HIP graph): an unselected expert's weights
are never read and its output counts as zero, which is what the multiplication by zero yields.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .._C import ops
from ..parallel_utils.communication_op import tensor_model_parallel_all_reduce
from ..parallel_utils.parallel_state import (get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size)
from .linear import LinearMethodBase, ReplicatedLinear, write_epoch
from .quantization.awq import AWQLinearMethod


class MixtralMLP(torch.nn.Module):
    """One expert: w2(silu(w1 x) * w3 x) (mixtral_quant.py:57-93).  Used on its own it runs three
    single launches through the layer's linear method, like the reference."""

    def __init__(self, num_experts: int, hidden_size: int, intermediate_size: int,
                 linear_method: Optional[LinearMethodBase] = None, params_dtype: Optional[torch.dtype] = None):
        super().__init__()
        self.num_experts = num_experts
        self.ffn_dim = intermediate_size
        self.hidden_dim = hidden_size
        self.w1 = ReplicatedLinear(hidden_size, intermediate_size, bias=False, params_dtype=params_dtype,
                                   linear_method=linear_method)
        self.w2 = ReplicatedLinear(intermediate_size, hidden_size, bias=False, params_dtype=params_dtype,
                                   linear_method=linear_method)
        self.w3 = ReplicatedLinear(hidden_size, intermediate_size, bias=False, params_dtype=params_dtype,
                                   linear_method=linear_method)

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        w1_out, _ = self.w1(hidden_states)
        w3_out, _ = self.w3(hidden_states)
        out, _ = self.w2(F.silu(w1_out) * w3_out)
        return out


class MixtralMoE(torch.nn.Module):
    """Router + this rank's experts (mixtral_quant.py:96-160).

    Args mirror the reference's (`config.num_local_experts`, `num_experts_per_tok`, `hidden_size`,
    `intermediate_size`) as plain integers, since no HF config class is needed for them.
    """

    def __init__(self, num_total_experts: int, top_k: int, hidden_size: int, intermediate_size: int,
                 linear_method: Optional[LinearMethodBase] = None, params_dtype: Optional[torch.dtype] = None):
        super().__init__()
        self.rank = get_tensor_model_parallel_rank()
        self.tp_size = get_tensor_model_parallel_world_size()
        self.num_total_experts = num_total_experts
        self.top_k = top_k
        if self.tp_size > num_total_experts:
            raise ValueError(f"Tensor parallel size {self.tp_size} is greater than "
                             f"the number of experts {num_total_experts}.")
        # experts are split equally between ranks (mixtral_quant.py:113-118)
        self.expert_indicies: List[int] = np.array_split(range(num_total_experts), self.tp_size)[self.rank].tolist()
        if not self.expert_indicies:
            raise ValueError(f"Rank {self.rank} has no experts assigned to it.")
        self.experts = torch.nn.ModuleList([
            MixtralMLP(num_total_experts, hidden_size, intermediate_size, linear_method, params_dtype)
            if idx in self.expert_indicies else None for idx in range(num_total_experts)
        ])
        self.gate = ReplicatedLinear(hidden_size, num_total_experts, bias=False, params_dtype=params_dtype,
                                     linear_method=None)
        self.batched = isinstance(linear_method, AWQLinearMethod)
        self.fuse_act = True  # the intermediate size is a multiple of the group size, hence of 16
        # one token selects top_k experts; 4 tokens of a top-2-of-8 router still leave a third of them idle,
        # beyond that the switch costs about what it saves
        self.sparse_max_tokens = 4
        # routing weights and the weighted sum over experts as one launch each instead of fourteen framework
        # kernels (GPU tensors only; the framework-operator route below stays for everything else)
        self.fused_routing = True
        self._packed = None  # (quick stamp, w13 batch, w2 batch, full stamp)
        self._watched: List[torch.Tensor] = []

    # ------------------------------------------------------------------
    def _stock(self, names):
        out = []
        for name in names:
            for idx in self.expert_indicies:
                lin = getattr(self.experts[idx], name)
                out.append((lin.qweight, lin.scales, lin.qzeros))
        return out

    def invalidate(self) -> None:
        """Forget the packed copy (after a write to an expert's tensors that neither went through a weight
        loader, nor bumped the tensor's in-place-write counter, nor moved its storage: `param.data.copy_(..)`,
        an in-place write to an inference-mode tensor)."""
        self._packed = None

    def _versions(self):
        """Address and in-place-write counter of every watched tensor: a storage swap (`param.data = ..`,
        `module.to(..)`) changes the first, a counted in-place write the second."""
        try:
            return tuple((t.data_ptr(), t._version) for t in self._watched)
        except RuntimeError:  # inference tensors keep no counter
            return tuple((t.data_ptr(), 0) for t in self._watched)

    def _batches(self):
        """The local experts' weights back to back in the private layout, rebuilt when any stock
        tensor was written since (weight loading) or moved.  The common case costs one integer comparison
        and one pass over the watched tensors' addresses and write counters: the experts' modules are walked
        again only after a weight loader has run or one of those changed."""
        if self._packed is not None and self._packed[0] == (write_epoch(), self.fuse_act, self._versions()):
            return self._packed[1], self._packed[2]
        w1, w3, w2 = self._stock(("w1", )), self._stock(("w3", )), self._stock(("w2", ))
        flat = [t for trio in w1 + w3 + w2 for t in trio]
        stamp = tuple((t.data_ptr(), ops.tensor_version(t)) for t in flat)
        if self._packed is not None and self._packed[3] == (stamp, self.fuse_act):
            self._watched = flat
            self._packed = ((write_epoch(), self.fuse_act, self._versions()), ) + self._packed[1:]
            return self._packed[1], self._packed[2]
        if ops.stream_is_capturing():
            return None  # no re-layout inside a graph capture
        if self.fuse_act:
            # one [K, 2 I] problem per expert: w1's columns, then w3's (all three AWQ tensors are
            # column-major in the output dimension, so this is a concatenation along dim 1)
            merged = [tuple(torch.cat([a, b], dim=1) for a, b in zip(t1, t3)) for t1, t3 in zip(w1, w3)]
        else:
            merged = w1 + w3  # 2E problems: every w1, then every w3
        b13 = ops.awq_repack_batch(*[[trio[i] for trio in merged] for i in (0, 1, 2)])
        b2 = ops.awq_repack_batch(*[[trio[i] for trio in w2] for i in (0, 1, 2)])
        self._watched = flat
        self._packed = ((write_epoch(), self.fuse_act, self._versions()), b13, b2, (stamp, self.fuse_act))
        return b13, b2

    def _forward_fused(self, hidden_states: torch.Tensor, b13, b2) -> torch.Tensor:
        """The block in four launches beside the gate: routing weights and problem mask, gate/up GEMM with the
        activation, down GEMM, weighted sum over experts.  Same arithmetic as the framework-operator route in
        `forward` (fp32 softmax / top-k / renormalise; each expert's output scaled in fp32 and rounded to fp16,
        the sum over experts in fp32 and rounded once), so one module gives the same values whichever route a
        call takes, up to the last bit of the fp32 routing weights (this kernel's `expf` and summation order
        against the framework's)."""
        router_logits, _ = self.gate(hidden_states)
        sparse = hidden_states.shape[0] <= self.sparse_max_tokens
        E = len(self.expert_indicies)
        weights, mask = ops.moe_route(router_logits, self.top_k, self.expert_indicies[0], E, want_mask=sparse)
        on = min(E, self.top_k * hidden_states.shape[0]) if sparse else 0
        # masked-out experts' rows are never read (their routing weight is 0 for every token): no fill needed
        act = ops.awq_gemm_packed_silu_and_mul(hidden_states, b13, mask=mask, active_hint=on, zero_masked=False)
        y = ops.awq_gemm_packed_batched(act, b2, mask=mask, active_hint=on, zero_masked=False)  # [E, T, H]
        return tensor_model_parallel_all_reduce(ops.moe_combine(y, weights))

    def routing(self, hidden_states: torch.Tensor) -> torch.Tensor:
        """[tokens, local experts] fp32 weights: softmax, top-k, renormalise (mixtral_quant.py:137-150)."""
        router_logits, _ = self.gate(hidden_states)
        routing_weights = F.softmax(router_logits, dim=1, dtype=torch.float)
        routing_weights, selected_experts = torch.topk(routing_weights, self.top_k, dim=-1)
        routing_weights /= routing_weights.sum(dim=-1, keepdim=True)
        dense = torch.zeros_like(router_logits, dtype=torch.float).scatter_(1, selected_experts, routing_weights)
        # the local experts are a contiguous range (np.array_split of a range): a slice, not an index
        # tensor, keeps this capturable in a HIP graph
        return dense[:, self.expert_indicies[0]:self.expert_indicies[-1] + 1]

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        batch_size, sequence_length, hidden_dim = hidden_states.shape
        hidden_states = hidden_states.view(-1, hidden_dim)
        batches = self._batches() if self.batched else None
        if batches is not None and self.fused_routing and self.fuse_act and hidden_states.is_cuda:
            return self._forward_fused(hidden_states, *batches).view(batch_size, sequence_length, hidden_dim)
        expert_weights = self.routing(hidden_states)  # [T, E_local]
        if batches is None:
            # the reference's loop (mixtral_quant.py:145-157), with the sum over experts kept in fp32 and
            # rounded once like the batched route below, so that one module gives the same values whichever
            # route a call takes (the reference adds the fp16 products one by one)
            scaled = [self.experts[idx](hidden_states).mul_(expert_weights[:, j:j + 1])
                      for j, idx in enumerate(self.expert_indicies)]
            final = torch.stack(scaled).sum(0, dtype=torch.float32).to(hidden_states.dtype)
        else:
            b13, b2 = batches
            E = len(self.expert_indicies)
            mask, on = None, 0
            if hidden_states.shape[0] <= self.sparse_max_tokens:
                mask = (expert_weights != 0).any(0).to(torch.int32)  # [E_local]: some token routed here
                on = min(E, self.top_k * hidden_states.shape[0])  # at most; known without asking the device
            if self.fuse_act:
                act = ops.awq_gemm_packed_silu_and_mul(hidden_states, b13, mask=mask, active_hint=on)  # [E, T, I]
            else:
                h13 = ops.awq_gemm_packed_batched(hidden_states, b13, mask=None if mask is None else mask.repeat(2),
                                                  active_hint=2 * on)  # [2E, T, I]
                act = F.silu(h13[:E]) * h13[E:]
            y = ops.awq_gemm_packed_batched(act, b2, mask=mask, active_hint=on)  # [E, T, H]
            # each expert's output is scaled in fp32 and rounded to fp16 as `.mul_` does; the sum
            # over experts is taken in fp32 and rounded once
            # (fp16 x fp32 promotes to an fp32 product; `sum(dtype=)` widens on load: same arithmetic as
            # converting first, two launches fewer)
            scaled = torch.mul(y, expert_weights.t().unsqueeze(-1)).to(y.dtype)
            final = scaled.sum(0, dtype=torch.float32).to(y.dtype)
        return tensor_model_parallel_all_reduce(final).view(batch_size, sequence_length, hidden_dim)
