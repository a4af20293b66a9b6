"""Checkpoint adjacency of the W4A16 path: where AWQ tensors come from and how they reach the layers.

Host-side mirror of the pieces of vllm/model_executor/weight_utils.py the quantised linear layers depend on
(reference lines in each docstring), restricted to LOCAL checkpoints (nothing is ever downloaded):

  get_quant_config          quantisation config from a checkpoint directory          (:86-122)
  hf_model_weights_iterator (name, tensor) pairs of the *.safetensors files            (:203-261)
  default_weight_loader     whole-tensor copy for parameters without a loader          (:280-284)
  load_stacked_weights      the name mapping + shard dispatch of a model's `load_weights` (llama.py:346-387)
  write_awq_checkpoint      a synthetic checkpoint in the stock AutoAWQ "GEMM" format, for tests and
                            benchmarks where no real checkpoint can be fetched

On-disk contract (AutoAWQ GEMM format, docs/source/quantization/auto_awq.rst:31 of the reference): per
linear `<prefix>.qweight [K, N/8] int32`, `<prefix>.qzeros [K/G, N/8] int32`, `<prefix>.scales [K/G, N]
fp16`; `quant_config.json` (or `quantize_config.json`) with `w_bit`/`bits`, `q_group_size`/`group_size`,
`zero_point`.
"""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple, Union

import torch
from safetensors import safe_open
from safetensors.torch import save_file

from .layers.linear import mark_written
from .layers.quantization import QuantizationConfig, get_quantization_config


def get_quant_config(quantization: str, model_path: str,
                     hf_quantization_config: Optional[dict] = None) -> QuantizationConfig:
    """weight_utils.py:86-122.  The HF config's `quantization_config` wins when present; otherwise exactly
    one `*.json` of the directory must end in one of the method's config file names."""
    quant_cls = get_quantization_config(quantization)
    if hf_quantization_config is not None:
        return quant_cls.from_config(hf_quantization_config)
    if not os.path.isdir(model_path):
        raise ValueError(f"{model_path} is not a local directory (checkpoints are never downloaded here)")
    config_files = glob.glob(os.path.join(model_path, "*.json"))
    quant_config_files = [f for f in config_files if any(f.endswith(x) for x in quant_cls.get_config_filenames())]
    if len(quant_config_files) == 0:
        raise ValueError(f"Cannot find the config file for {quantization}")
    if len(quant_config_files) > 1:
        raise ValueError(f"Found multiple config files for {quantization}: {quant_config_files}")
    with open(quant_config_files[0], "r") as f:
        return quant_cls.from_config(json.load(f))


def hf_model_weights_iterator(model_path: str) -> Iterator[Tuple[str, torch.Tensor]]:
    """The safetensors branch of weight_utils.py:203-261: every tensor of every `*.safetensors` file of a
    local directory, in file then key order.  safetensors reads execute nothing from the file."""
    files = sorted(glob.glob(os.path.join(model_path, "*.safetensors")))
    if not files:
        raise RuntimeError(f"Cannot find any model weights with `{model_path}`")
    for st_file in files:
        with safe_open(st_file, framework="pt") as f:
            for name in f.keys():  # noqa: SIM118
                yield name, f.get_tensor(name)


def default_weight_loader(param: torch.Tensor, loaded_weight: torch.Tensor) -> None:
    """weight_utils.py:280-284, plus the write mark the W4A16 method's packed copy watches."""
    assert param.size() == loaded_weight.size()
    param.data.copy_(loaded_weight)
    mark_written(param)


# (param_name, shard_name, shard_id): llama.py:351-358
LLAMA_STACKED_PARAMS = [
    ("qkv_proj", "q_proj", "q"),
    ("qkv_proj", "k_proj", "k"),
    ("qkv_proj", "v_proj", "v"),
    ("gate_up_proj", "gate_proj", 0),
    ("gate_up_proj", "up_proj", 1),
]


def load_stacked_weights(module: torch.nn.Module, weights: Iterable[Tuple[str, torch.Tensor]],
                         stacked_params_mapping: Sequence[Tuple[str, str, Union[str, int]]] = tuple(LLAMA_STACKED_PARAMS),
                         strict: bool = True) -> List[str]:
    """The body of a model's `load_weights` (llama.py:359-387): checkpoint tensors named after the
    un-fused projections are routed to the fused parameters with their shard id; everything else goes
    to the parameter of the same name through its `weight_loader` (or the default one).  Returns the
    names it loaded.  With `strict=False` tensors without a parameter are skipped instead of raising."""
    params_dict = dict(module.named_parameters())
    loaded = []
    for name, loaded_weight in weights:
        if "rotary_emb.inv_freq" in name or "rotary_emb.cos_cached" in name or "rotary_emb.sin_cached" in name:
            continue
        for param_name, weight_name, shard_id in stacked_params_mapping:
            if weight_name not in name:
                continue
            target = name.replace(weight_name, param_name)
            if target.endswith(".bias") and target not in params_dict:  # extra GPTQ biases (llama.py:373-375)
                break
            if target not in params_dict:
                if strict:
                    raise KeyError(f"checkpoint tensor {name} has no parameter {target}")
                break
            param = params_dict[target]
            param.weight_loader(param, loaded_weight, shard_id)
            loaded.append(name)
            break
        else:
            if name.endswith(".bias") and name not in params_dict:
                continue
            if name not in params_dict:
                if strict:
                    raise KeyError(f"checkpoint tensor {name} has no parameter")
                continue
            param = params_dict[name]
            getattr(param, "weight_loader", default_weight_loader)(param, loaded_weight)
            loaded.append(name)
    return loaded


def synthetic_awq_linear(in_features: int, out_features: int, group_size: int,
                         generator: torch.Generator) -> Dict[str, torch.Tensor]:
    """Seeded stock-format tensors of one linear: every 32-bit pattern equally likely (so every nibble
    value occurs), scales U(1e-3, 2e-2)."""
    def words(rows):
        return torch.randint(-2**31, 2**31 - 1, (rows, out_features // 8), generator=generator,
                             dtype=torch.int64).to(torch.int32)

    groups = in_features // group_size
    return {
        "qweight": words(in_features),
        "qzeros": words(groups),
        "scales": (torch.rand((groups, out_features), generator=generator) * (2e-2 - 1e-3) + 1e-3).half(),
    }


def write_awq_checkpoint(path: str, linears: Dict[str, Tuple[int, int]], group_size: int = 128, seed: int = 0,
                         extra: Optional[Dict[str, torch.Tensor]] = None,
                         config_name: str = "quant_config.json", shard_bytes: int = 1 << 30) -> List[str]:
    """Writes `<path>/model-XXXXX.safetensors` + `<path>/<config_name>` for the given linears
    ({prefix: (in_features, out_features)}) in the stock AutoAWQ GEMM format.  Returns the file names."""
    os.makedirs(path, exist_ok=True)
    gen = torch.Generator()
    gen.manual_seed(seed)
    files, cur, cur_bytes = [], {}, 0

    def flush():
        nonlocal cur, cur_bytes
        if cur:
            name = os.path.join(path, f"model-{len(files) + 1:05d}.safetensors")
            save_file(cur, name)
            files.append(name)
            cur, cur_bytes = {}, 0

    for prefix, (k, n) in linears.items():
        if k % group_size or n % 8:
            raise ValueError(f"{prefix}: {k}x{n} does not fit group size {group_size} / pack factor 8")
        for suffix, t in synthetic_awq_linear(k, n, group_size, gen).items():
            cur[f"{prefix}.{suffix}"] = t
            cur_bytes += t.numel() * t.element_size()
        if cur_bytes >= shard_bytes:
            flush()
    for name, t in (extra or {}).items():
        cur[name] = t.contiguous()
    flush()
    with open(os.path.join(path, config_name), "w") as f:
        json.dump({"w_bit": 4, "q_group_size": group_size, "zero_point": True, "version": "GEMM"}, f)
    return files
