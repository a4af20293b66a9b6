"""Abstract quantisation config: the plugin surface model loaders program against.

Mirror of vllm/model_executor/layers/quantization/base_config.py:9-64.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Dict, List

import torch

from ..linear import LinearMethodBase


class QuantizationConfig(ABC):

    @abstractmethod
    def get_name(self) -> str:
        """Registry name of the method (e.g. "awq")."""

    @abstractmethod
    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        """Activation dtypes the kernels accept."""

    @abstractmethod
    def get_min_capability(self) -> int:
        """Minimum CUDA compute capability the reference kernels needed (kept for API parity)."""

    @staticmethod
    @abstractmethod
    def get_config_filenames() -> List[str]:
        """Files to look for in a checkpoint directory."""

    @classmethod
    @abstractmethod
    def from_config(cls, config: Dict[str, Any]) -> "QuantizationConfig":
        """Build from the checkpoint's quantisation JSON."""

    @staticmethod
    def get_from_keys(config: Dict[str, Any], keys: List[str]) -> Any:
        """First of `keys` present in `config`; ValueError if none is."""
        for key in keys:
            if key in config:
                return config[key]
        raise ValueError(f"Cannot find any of {keys} in the model's quantization config.")

    @abstractmethod
    def get_linear_method(self) -> LinearMethodBase:
        """The LinearMethodBase that implements this quantisation."""

    @abstractmethod
    def get_scaled_act_names(self) -> List[str]:
        """Activation names that need a post-scale (AWQ only)."""
