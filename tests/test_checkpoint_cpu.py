"""Checkpoint adjacency without a GPU: a synthetic checkpoint in the stock AutoAWQ GEMM format is written,
its quantisation config found the way the reference finds it (weight_utils.py:86-122), and its tensors routed
into fused, tensor-parallel layers by the reference's name mapping (llama.py:346-387); the result must equal
slicing the checkpoint tensors by hand (linear.py:191-201, 272-281, 389-397, 516-526)."""
import json
import os

import numpy as np
import pytest
import torch

from vllm_quick_amd import weight_utils as WU
from vllm_quick_amd.layers import linear as L
from vllm_quick_amd.layers.quantization import AWQConfig

H, I, HEADS, KV, HD, G = 128, 256, 4, 2, 32, 32


class Attn(torch.nn.Module):

    def __init__(self, method):
        super().__init__()
        self.qkv_proj = L.QKVParallelLinear(H, HD, HEADS, KV, bias=False, params_dtype=torch.float16,
                                            linear_method=method)
        self.o_proj = L.RowParallelLinear(HEADS * HD, H, bias=False, params_dtype=torch.float16,
                                          linear_method=method)


class MLP(torch.nn.Module):

    def __init__(self, method):
        super().__init__()
        self.gate_up_proj = L.MergedColumnParallelLinear(H, [I, I], bias=False, params_dtype=torch.float16,
                                                         linear_method=method)
        self.down_proj = L.RowParallelLinear(I, H, bias=False, params_dtype=torch.float16, linear_method=method)


class Layer(torch.nn.Module):

    def __init__(self, method):
        super().__init__()
        self.self_attn = Attn(method)
        self.mlp = MLP(method)
        self.input_layernorm = torch.nn.Module()
        self.input_layernorm.weight = torch.nn.Parameter(torch.ones(H, dtype=torch.float16), requires_grad=False)


class Model(torch.nn.Module):

    def __init__(self, method, n_layers=2):
        super().__init__()
        self.layers = torch.nn.ModuleList([Layer(method) for _ in range(n_layers)])


def linears():
    out = {}
    for i in range(2):
        p = f"layers.{i}"
        out[f"{p}.self_attn.q_proj"] = (H, HEADS * HD)
        out[f"{p}.self_attn.k_proj"] = (H, KV * HD)
        out[f"{p}.self_attn.v_proj"] = (H, KV * HD)
        out[f"{p}.self_attn.o_proj"] = (HEADS * HD, H)
        out[f"{p}.mlp.gate_proj"] = (H, I)
        out[f"{p}.mlp.up_proj"] = (H, I)
        out[f"{p}.mlp.down_proj"] = (I, H)
    return out


@pytest.fixture()
def ckpt_dir(tmp_path):
    extra = {f"layers.{i}.input_layernorm.weight": torch.full((H, ), 1.0 + i, dtype=torch.float16) for i in range(2)}
    extra["layers.0.self_attn.rotary_emb.inv_freq"] = torch.ones(4)
    WU.write_awq_checkpoint(str(tmp_path), linears(), group_size=G, seed=5, extra=extra, shard_bytes=64 * 1024)
    return str(tmp_path)


def test_quant_config_discovery(ckpt_dir, tmp_path):
    cfg = WU.get_quant_config("awq", ckpt_dir)
    assert isinstance(cfg, AWQConfig) and (cfg.weight_bits, cfg.group_size, cfg.zero_point) == (4, G, True)
    # the HF config's own section wins (weight_utils.py:89-93)
    cfg2 = WU.get_quant_config("awq", ckpt_dir, {"bits": 4, "group_size": 64, "zero_point": True})
    assert cfg2.group_size == 64
    with open(os.path.join(ckpt_dir, "quantize_config.json"), "w") as f:
        json.dump({"bits": 4, "group_size": G, "zero_point": True}, f)
    with pytest.raises(ValueError, match="multiple config files"):
        WU.get_quant_config("awq", ckpt_dir)
    empty = tmp_path / "empty"
    empty.mkdir()
    with pytest.raises(ValueError, match="Cannot find the config file"):
        WU.get_quant_config("awq", str(empty))
    with pytest.raises(RuntimeError, match="Cannot find any model weights"):
        list(WU.hf_model_weights_iterator(str(empty)))


def by_hand(ck, rank, tp):
    """What every parameter of rank `rank` must hold, from the checkpoint tensors directly."""
    want = {}

    def cols(t, lo, hi, packed):
        return t[:, lo // 8:hi // 8] if packed else t[:, lo:hi]

    for i in range(2):
        p = f"layers.{i}"
        for suffix in ("qweight", "qzeros", "scales"):
            packed = suffix != "scales"
            q, k, v = (ck[f"{p}.self_attn.{n}_proj.{suffix}"] for n in "qkv")
            qw, kw = HEADS * HD // tp, KV * HD // tp
            want[f"{p}.self_attn.qkv_proj.{suffix}"] = torch.cat(
                [cols(q, rank * qw, (rank + 1) * qw, packed), cols(k, rank * kw, (rank + 1) * kw, packed),
                 cols(v, rank * kw, (rank + 1) * kw, packed)], 1)
            g, u = ck[f"{p}.mlp.gate_proj.{suffix}"], ck[f"{p}.mlp.up_proj.{suffix}"]
            iw = I // tp
            want[f"{p}.mlp.gate_up_proj.{suffix}"] = torch.cat(
                [cols(g, rank * iw, (rank + 1) * iw, packed), cols(u, rank * iw, (rank + 1) * iw, packed)], 1)
            for name in ("self_attn.o_proj", "mlp.down_proj"):
                t = ck[f"{p}.{name}.{suffix}"]
                rows = t.shape[0] // tp
                want[f"{p}.{name}.{suffix}"] = t[rank * rows:(rank + 1) * rows]
        want[f"{p}.input_layernorm.weight"] = ck[f"{p}.input_layernorm.weight"]
    return want


@pytest.mark.parametrize("tp,rank", [(1, 0), (2, 0), (2, 1)])
def test_load_stacked_weights_matches_slicing_by_hand(ckpt_dir, monkeypatch, tp, rank):
    monkeypatch.setattr(L, "get_tensor_model_parallel_world_size", lambda: tp)
    monkeypatch.setattr(L, "get_tensor_model_parallel_rank", lambda: rank)
    method = WU.get_quant_config("awq", ckpt_dir).get_linear_method()
    model = Model(method)
    ck = dict(WU.hf_model_weights_iterator(ckpt_dir))
    assert len([f for f in os.listdir(ckpt_dir) if f.endswith(".safetensors")]) > 1  # sharded on disk
    loaded = WU.load_stacked_weights(model, WU.hf_model_weights_iterator(ckpt_dir))
    assert "layers.0.self_attn.rotary_emb.inv_freq" not in loaded and len(loaded) == len(ck) - 1
    params = dict(model.named_parameters())
    want = by_hand(ck, rank, tp)
    assert set(want) == set(params)
    for name, t in want.items():
        assert torch.equal(params[name].data, t), name
    # every loaded parameter carries the write mark the packed copies watch
    assert all(p._version > 0 for n, p in params.items())
    with pytest.raises(KeyError):
        WU.load_stacked_weights(model, [("layers.9.mlp.down_proj.qweight", torch.zeros(1))])
    assert WU.load_stacked_weights(model, [("layers.9.mlp.down_proj.qweight", torch.zeros(1))], strict=False) == []


def test_a_loader_call_bumps_the_write_counter_by_exactly_one():
    # the packed copy of a layer's weights is keyed on this counter: one bump per load, not one per row
    p = torch.nn.Parameter(torch.zeros(4, 3), requires_grad=False)
    before = p._version
    WU.default_weight_loader(p, torch.ones(4, 3))
    assert p._version == before + 1 and bool((p == 1).all())
    L.mark_written(p)
    assert p._version == before + 2
    # a 0-d parameter (no rows to walk) loads too
    s = torch.nn.Parameter(torch.zeros(()), requires_grad=False)
    before = s._version
    WU.default_weight_loader(s, torch.tensor(2.5))
    assert s._version == before + 1 and float(s) == 2.5
