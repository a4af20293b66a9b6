"""Quantisation registry (reference: vllm/model_executor/layers/quantization/__init__.py:8-18).

Only "awq" is provided: it is the one method whose kernels this package implements.
"""
from typing import Type

from .awq import AWQConfig
from .base_config import QuantizationConfig

_QUANTIZATION_CONFIG_REGISTRY = {
    "awq": AWQConfig,
}


def get_quantization_config(quantization: str) -> Type[QuantizationConfig]:
    if quantization not in _QUANTIZATION_CONFIG_REGISTRY:
        raise ValueError(f"Invalid quantization method: {quantization}")
    return _QUANTIZATION_CONFIG_REGISTRY[quantization]


__all__ = ["QuantizationConfig", "get_quantization_config", "AWQConfig"]
