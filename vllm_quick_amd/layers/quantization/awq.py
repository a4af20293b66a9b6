"""AWQ (4-bit weights, fp16 activations) linear method on the MI355X kernels.

Mirror of vllm/model_executor/layers/quantization/awq.py: `AWQConfig` (:12-67)
and `AWQLinearMethod.create_weights` (:80-141) / `apply_weights` (:143-165) keep
the reference's parameter names, shapes, dtypes, loader attributes, error
conditions and the M >= 256 dequantise-then-matmul dispatch.  What differs is
underneath: the first `apply_weights` on a layer re-orders its weights once into
libw4a16's MFMA-ordered layout (kept in the layer's weights dict next to the
stock tensors, the way the reference's GPTQ method keeps `exllama_state`,
gptq.py:180-186) and later calls go straight to the packed-layout kernel, with
the bias add folded into its epilogue.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import torch
from torch.nn.parameter import Parameter

from ..._C import ops
from ..linear import LinearMethodBase, set_weight_attrs
from .base_config import QuantizationConfig

_PACKED_KEY = "w4a16_packed"
_DROPPED_KEY = "w4a16_stock_dropped"


class AWQConfig(QuantizationConfig):
    """Config class for AWQ (https://arxiv.org/abs/2306.00978)."""

    def __init__(self, weight_bits: int, group_size: int, zero_point: bool) -> None:
        self.weight_bits = weight_bits
        self.group_size = group_size
        self.zero_point = zero_point
        if self.weight_bits != 4:
            raise ValueError("Currently, only 4-bit weight quantization is supported for "
                             f"AWQ, but got {self.weight_bits} bits.")
        self.pack_factor = 32 // self.weight_bits

    def __repr__(self) -> str:
        return (f"AWQConfig(weight_bits={self.weight_bits}, "
                f"group_size={self.group_size}, "
                f"zero_point={self.zero_point})")

    def get_name(self) -> str:
        return "awq"

    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        return [torch.half]

    def get_min_capability(self) -> int:
        # Meaningful for the reference's CUDA kernel only (Turing or newer); kept for API parity.
        return 75

    @staticmethod
    def get_config_filenames() -> List[str]:
        return ["quant_config.json", "quantize_config.json"]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "AWQConfig":
        weight_bits = cls.get_from_keys(config, ["w_bit", "bits"])
        group_size = cls.get_from_keys(config, ["q_group_size", "group_size"])
        zero_point = cls.get_from_keys(config, ["zero_point"])
        return cls(weight_bits, group_size, zero_point)

    def get_linear_method(self) -> "AWQLinearMethod":
        return AWQLinearMethod(self)

    def get_scaled_act_names(self) -> List[str]:
        return ["gelu", "gelu_fast", "gelu_new", "gelu_pytorch_tanh"]


class AWQLinearMethod(LinearMethodBase):
    """Linear method for AWQ.

    Args:
        quant_config: The AWQ quantization config.
        fp16_matmul_threshold: token count from which `apply_weights` dequantises the
            weights to HBM and calls `torch.matmul`, the reference's large-batch route
            (awq.py:154-159).  The reference switches at 256 tokens, a heuristic tuned to its
            CUDA kernel; on MI355X the fused kernels stay ahead of dequantise + hipBLASLt at every
            token count measured (2.5x at 256 tokens, 1.2x at 1024, 1.1x at 4096 on 4096 x 11008), so
            the default here is `None`: always the fused kernels.  Pass 256 for the reference's exact
            operator choice.
        keep_stock: `False` lets `drop_stock()` release a layer's stock tensors once they have been
            re-ordered into the private layout (the reference's GPTQ path re-orders its weights in place
            on first use in the same spirit, gptq.py:197-204); otherwise both copies stay resident.
        split_k: up to 4 tokens, let the library split K over workgroups where a layer has too few
            columns to fill the chip (a tensor-parallel rank's qkv projection), through one workspace per
            stream (`ops.stream_workspace`).  The reference always splits K eight ways through a
            [split_k, M, N] tensor (gemm_kernels.cu:420-445).  Off by default: launches that share a
            workspace must not overlap, which holds for everything enqueued on one stream but is the
            caller's business once graphs captured on one stream are replayed on several.
    """

    REFERENCE_FP16_MATMUL_THRESHOLD = 256
    DEFAULT_FP16_MATMUL_THRESHOLD = None

    def __init__(self, quant_config: AWQConfig,
                 fp16_matmul_threshold: Optional[int] = DEFAULT_FP16_MATMUL_THRESHOLD, keep_stock: bool = True,
                 split_k: bool = False):
        self.quant_config = quant_config
        self.fp16_matmul_threshold = fp16_matmul_threshold
        self.keep_stock = keep_stock
        self.split_k = split_k

    def create_weights(self, input_size_per_partition: int, output_size_per_partition: int, input_size: int,
                       output_size: int, params_dtype: torch.dtype) -> Dict[str, Any]:
        group, pack = self.quant_config.group_size, self.quant_config.pack_factor
        misaligned = ("The {} size is not aligned with the quantized weight shape. "
                      "This can be caused by too large tensor parallel size.")
        if input_size_per_partition % group != 0:
            raise ValueError(misaligned.format("input"))
        if output_size_per_partition % pack != 0:
            raise ValueError(misaligned.format("output"))

        def packed_param(rows: int) -> Parameter:
            p = Parameter(torch.empty(rows, output_size_per_partition // pack, dtype=torch.int32),
                          requires_grad=False)
            set_weight_attrs(p, {"input_dim": 0, "output_dim": 1, "packed_dim": 1, "pack_factor": pack})
            return p

        qweight = packed_param(input_size_per_partition)
        qzeros = packed_param(input_size_per_partition // group)
        scales = Parameter(torch.empty(input_size_per_partition // group, output_size_per_partition,
                                       dtype=params_dtype),
                           requires_grad=False)
        set_weight_attrs(scales, {"input_dim": 0, "output_dim": 1})
        return {"qweight": qweight, "qzeros": qzeros, "scales": scales}

    def drop_stock(self, weights: Dict[str, Any]) -> int:
        """Re-orders the layer's weights into the private layout (if that has not happened yet) and
        releases the stock `qweight` / `qzeros` / `scales` storages: from then on the layer holds its
        4-bit weights once instead of twice.  Call after weight loading, outside graph capture.  The
        parameters stay registered (empty) so that module traversal keeps working; loading new weights
        into them afterwards is not supported.  Returns the bytes released (0 if `keep_stock`)."""
        if self.keep_stock or weights.get(_DROPPED_KEY):
            return 0
        packed = self._packed(weights)
        if packed is None:
            raise RuntimeError("drop_stock() cannot re-order weights during graph capture")
        freed = 0
        for name in ("qweight", "qzeros", "scales"):
            t = weights[name]
            freed += t.numel() * t.element_size()
            t.data = torch.empty(0, dtype=t.dtype, device=t.device)
        weights[_DROPPED_KEY] = True
        return freed

    @staticmethod
    def _packed(weights: Dict[str, Any]) -> Optional[ops.PackedAWQ]:
        """The layer's MFMA-ordered weights, (re)built when the stock tensors changed."""
        if weights.get(_DROPPED_KEY):
            return weights[_PACKED_KEY][1]
        stock = (weights["qweight"], weights["scales"], weights["qzeros"])
        stamp = tuple((t.data_ptr(), ops.tensor_version(t)) for t in stock)
        state = weights.get(_PACKED_KEY)
        if state is not None and state[0] == stamp:
            return state[1]
        if ops.stream_is_capturing():
            return None  # no re-layout inside a graph capture; the stock-layout kernel serves this call
        packed = ops.awq_repack(*stock)
        weights[_PACKED_KEY] = (stamp, packed)
        return packed

    def apply_weights(self, weights: Dict[str, Any], x: torch.Tensor,
                      bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        qweight, scales, qzeros = weights["qweight"], weights["scales"], weights["qzeros"]
        dropped = bool(weights.get(_DROPPED_KEY))
        n_out = weights[_PACKED_KEY][1].N if dropped else qweight.shape[-1] * self.quant_config.pack_factor
        out_shape = x.shape[:-1] + (n_out, )
        reshaped_x = x.reshape(-1, x.shape[-1])
        num_tokens = x.shape[:-1].numel()

        if self.fp16_matmul_threshold is not None and num_tokens >= self.fp16_matmul_threshold and x.dtype == torch.float16:
            # (bit-identical dense weights from either layout; fp16 only: there is no bf16 dequantise)
            dense = ops.awq_dequantize_packed(weights[_PACKED_KEY][1]) if dropped else \
                ops.awq_dequantize(qweight, scales, qzeros, 0, 0, 0)
            out = torch.matmul(reshaped_x, dense)
            if bias is not None:
                out = out + bias
            return out.reshape(out_shape)

        packed = self._packed(weights)
        if packed is None:
            out = ops.awq_gemm_stock(reshaped_x, qweight, scales, qzeros)
            if bias is not None:
                out = out + bias
        else:
            if self.split_k and num_tokens <= 4:
                out = ops.awq_gemm_packed(reshaped_x, packed, bias=bias,
                                          workspace=ops.stream_workspace(reshaped_x.device))
            else:
                out = ops.awq_gemm_packed(reshaped_x, packed, bias=bias)
        return out.reshape(out_shape)

    def apply_weights_silu_and_mul(self, weights: Dict[str, Any], x: torch.Tensor,
                                   bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        """silu(y[..., :N/2]) * y[..., N/2:] for y = apply_weights(weights, x, bias): the merged gate_up
        projection of a gated MLP followed by `SiluAndMul` (llama.py:74-80), with the activation fused
        into the GEMM epilogue where the fused kernel serves the call."""
        n_out = weights[_PACKED_KEY][1].N if weights.get(_DROPPED_KEY) else \
            weights["qweight"].shape[-1] * self.quant_config.pack_factor
        num_tokens = x.shape[:-1].numel()
        large = self.fp16_matmul_threshold is not None and num_tokens >= self.fp16_matmul_threshold
        # (the fused activation is built for fp16; bf16 activations take the two separate steps)
        packed = None if (large or n_out % 32 or x.dtype != torch.float16) else self._packed(weights)
        if packed is None:
            y = self.apply_weights(weights, x, bias)
            half = n_out // 2
            return torch.nn.functional.silu(y[..., :half]) * y[..., half:]
        out = ops.awq_gemm_packed_silu_and_mul(x.reshape(-1, x.shape[-1]), packed, bias=bias)
        return out.reshape(x.shape[:-1] + (n_out // 2, ))

    def apply_weights_add_(self, weights: Dict[str, Any], x: torch.Tensor, residual: torch.Tensor,
                           bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        """residual += apply_weights(weights, x, bias), in place; returns `residual`.  The `residual +
        o_proj(...)` / `residual + down_proj(...)` of a decoder layer (llama.py:197-223) with the add in the
        GEMM epilogue where the fused kernel serves the call; same bits as the two separate steps."""
        num_tokens = x.shape[:-1].numel()
        large = self.fp16_matmul_threshold is not None and num_tokens >= self.fp16_matmul_threshold
        packed = None if large else self._packed(weights)
        if packed is not None and residual.is_contiguous() and residual.dtype == torch.bfloat16 and x.dtype == torch.bfloat16:
            ops.awq_gemm_packed_bf16(x.reshape(-1, x.shape[-1]), packed, bias=bias, out=residual.view(-1, residual.shape[-1]))
            return residual
        if packed is None or not residual.is_contiguous() or residual.dtype != torch.float16 or x.dtype != torch.float16:
            return residual.add_(self.apply_weights(weights, x, bias))
        ops.awq_gemm_packed_add_(residual.view(-1, residual.shape[-1]), x.reshape(-1, x.shape[-1]), packed, bias=bias)
        return residual
