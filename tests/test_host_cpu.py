"""The host-side mirror (AWQConfig, AWQLinearMethod, linear layers) against behaviour captured
from the REFERENCE's own Python (tests/golden/awq_reference_python.*, made by
oracle/gen_golden.py, which imported /root/reference with the native ops faked by the oracle).

No GPU: the native operators are replaced here by the same oracle-backed fakes, so what is
compared is everything around the kernels - parameters, attributes, errors, dispatch, bias,
reshape and tensor-parallel shard arithmetic.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import awq_oracle as orc
from _fakes import FakeOps
from vllm_quick_amd.layers import linear as L
from vllm_quick_amd.layers.quantization import AWQConfig, QuantizationConfig, get_quantization_config
from vllm_quick_amd.layers.quantization import awq as awq_mod

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def meta():
    return json.load(open(os.path.join(G, "awq_reference_python.json")))


@pytest.fixture(scope="module")
def arrays():
    return np.load(os.path.join(G, "awq_reference_python.npz"))


def caught(fn):
    try:
        fn()
        return None
    except Exception as e:  # noqa: BLE001
        return [type(e).__name__, str(e)]


def test_config_surface(meta):
    want = meta["config"]
    cfg = AWQConfig.from_config({"w_bit": 4, "q_group_size": 128, "zero_point": True})
    cfg2 = AWQConfig.from_config({"bits": 4, "group_size": 64, "zero_point": False})
    assert repr(cfg) == want["repr"] and repr(cfg2) == want["repr2"]
    assert cfg.get_name() == want["name"]
    assert cfg.pack_factor == want["pack_factor"]
    assert [str(d) for d in cfg.get_supported_act_dtypes()] == want["act_dtypes"]
    assert cfg.get_min_capability() == want["min_capability"]
    assert AWQConfig.get_config_filenames() == want["filenames"]
    assert cfg.get_scaled_act_names() == want["scaled_act_names"]
    assert (get_quantization_config("awq") is AWQConfig) == want["registry_awq_is_AWQConfig"]
    assert isinstance(cfg, QuantizationConfig)
    assert isinstance(cfg.get_linear_method(), L.LinearMethodBase)


def test_config_errors(meta):
    want = meta["config_errors"]
    assert caught(lambda: AWQConfig(8, 128, True)) == want["bits8"]
    assert caught(lambda: AWQConfig.from_config({"bits": 4, "zero_point": True})) == want["missing_key"]
    assert caught(lambda: get_quantization_config("nope")) == want["bad_registry"]


def test_create_weights(meta):
    method = AWQConfig(4, 128, True).get_linear_method()
    w = method.create_weights(4096, 11008, 4096, 11008, torch.float16)
    want = meta["create_weights_4096x11008"]
    assert sorted(w) == sorted(want)
    for name, spec in want.items():
        p = w[name]
        assert isinstance(p, torch.nn.Parameter)
        assert list(p.shape) == spec["shape"] and str(p.dtype) == spec["dtype"]
        assert p.requires_grad == spec["requires_grad"]
        got = {k: getattr(p, k) for k in ("input_dim", "output_dim", "packed_dim", "pack_factor") if hasattr(p, k)}
        assert got == spec["attrs"]
    errs = meta["create_weights_errors"]
    assert caught(lambda: method.create_weights(4096, 11004, 4096, 11004, torch.float16)) == errs["n_unaligned"]
    assert caught(lambda: method.create_weights(4000, 11008, 4000, 11008, torch.float16)) == errs["k_unaligned"]


def test_apply_weights_dispatch_and_values(meta, arrays, monkeypatch):
    fake = FakeOps()
    monkeypatch.setattr(awq_mod, "ops", fake)
    # with the reference's switch point the operator choice must match its recorded trace
    method = awq_mod.AWQLinearMethod(AWQConfig(4, 128, True),
                                     fp16_matmul_threshold=awq_mod.AWQLinearMethod.REFERENCE_FP16_MATMUL_THRESHOLD)
    weights = {k: torch.from_numpy(arrays[f"apply/{k}"]) for k in ("qweight", "scales", "qzeros")}
    bias = torch.from_numpy(arrays["apply/bias"])
    for label, spec in meta["apply_weights"].items():
        x = torch.from_numpy(arrays[f"apply/{label}/x"])
        del fake.trace[:]
        y = method.apply_weights(weights, x, bias if spec["bias"] else None)
        assert list(y.shape) == spec["out_shape"] and str(y.dtype) == spec["out_dtype"], label
        ops_called = [t for t in fake.trace if t[0] != "repack"]
        ref = spec["trace"]
        assert len(ops_called) == len(ref) == 1, label
        assert ops_called[0][0] == ref[0][0], label  # same operator chosen at this token count
        if ref[0][0] == "awq_gemm":
            assert ops_called[0][1] == ref[0][1] and ops_called[0][2] == ref[0][3], label  # 2-D, contiguous
        else:
            assert ops_called[0][1:] == ref[0][1:], label  # (qweight shape, 0, 0, 0)
        want = arrays[f"apply/{label}/out"].astype(np.float32)
        assert np.allclose(y.numpy().astype(np.float32), want, rtol=1e-2, atol=1e-3), label
    # the re-layout happens once per layer, not once per call
    del fake.trace[:]
    x = torch.from_numpy(arrays["apply/m1/x"])
    method.apply_weights(weights, x)
    method.apply_weights(weights, x)
    assert [t[0] for t in fake.trace].count("repack") == 0
    weights["qweight"].add_(0)  # an in-place write (e.g. weight loading) invalidates the packed copy
    method.apply_weights(weights, x)
    assert [t[0] for t in fake.trace].count("repack") == 1


def test_threshold_none_always_uses_fused_kernel(arrays, monkeypatch):
    fake = FakeOps()
    monkeypatch.setattr(awq_mod, "ops", fake)
    method = awq_mod.AWQLinearMethod(AWQConfig(4, 128, True), fp16_matmul_threshold=None)
    weights = {k: torch.from_numpy(arrays[f"apply/{k}"]) for k in ("qweight", "scales", "qzeros")}
    method.apply_weights(weights, torch.from_numpy(arrays["apply/m256/x"]))
    assert [t[0] for t in fake.trace if t[0] != "repack"] == ["awq_gemm"]


def test_default_is_the_fused_kernel_at_every_batch(arrays, monkeypatch):
    """By default the fused kernels serve every batch (they are ahead of dequantise + matmul at every token
    count measured on MI355X); the reference's route at its 256-token switch point (awq.py:154-159) is one
    argument away."""
    fake = FakeOps()
    monkeypatch.setattr(awq_mod, "ops", fake)
    method = AWQConfig(4, 128, True).get_linear_method()
    assert method.fp16_matmul_threshold is None and method.keep_stock
    weights = {k: torch.from_numpy(arrays[f"apply/{k}"]) for k in ("qweight", "scales", "qzeros")}
    x = torch.from_numpy(arrays["apply/m256/x"])
    big = x.repeat(4, 1)
    y = method.apply_weights(weights, big)
    assert [t[0] for t in fake.trace if t[0] != "repack"] == ["awq_gemm"]
    want = arrays["apply/m256/out"].astype(np.float32)
    assert np.allclose(y[:256].numpy().astype(np.float32), want, rtol=1e-2, atol=1e-3)
    ref = awq_mod.AWQLinearMethod(AWQConfig(4, 128, True), fp16_matmul_threshold=256)
    ref.apply_weights(weights, x)
    assert fake.trace[-1][0] == "awq_dequantize"


def test_drop_stock_releases_the_stock_tensors(arrays, monkeypatch):
    """`keep_stock=False`: after drop_stock() the layer answers from the private layout alone, on the fused
    route and on the dense large-batch route (dequantised from the private layout, bit-identical)."""
    fake = FakeOps()
    monkeypatch.setattr(awq_mod, "ops", fake)
    weights = {k: torch.from_numpy(arrays[f"apply/{k}"]).clone() for k in ("qweight", "scales", "qzeros")}
    x = torch.from_numpy(arrays["apply/m256/x"])
    keep = AWQConfig(4, 128, True).get_linear_method()
    want = keep.apply_weights(dict(weights), x)
    assert keep.drop_stock(weights) == 0 and weights["qweight"].numel() > 0  # keep_stock: nothing released
    method = awq_mod.AWQLinearMethod(AWQConfig(4, 128, True), keep_stock=False)
    nbytes = sum(t.numel() * t.element_size() for t in weights.values())
    assert method.drop_stock(weights) == nbytes
    assert all(weights[k].numel() == 0 for k in ("qweight", "scales", "qzeros"))
    assert method.drop_stock(weights) == 0  # idempotent
    y = method.apply_weights(weights, x.view(2, 128, -1))
    assert y.shape == (2, 128, want.shape[-1]) and torch.equal(y.view(256, -1), want)
    dense = awq_mod.AWQLinearMethod(AWQConfig(4, 128, True), fp16_matmul_threshold=256, keep_stock=False)
    yd = dense.apply_weights(weights, x)
    assert fake.trace[-1][0] == "awq_dequantize_packed"
    assert np.allclose(yd.numpy().astype(np.float32), want.numpy().astype(np.float32), rtol=1e-2, atol=1e-3)


# --------------------------------------------------------------------------
# tensor-parallel weight loaders
# --------------------------------------------------------------------------
def ckpt(K, N, seed):
    _, qw, s, qz = orc.make_inputs(1, K, N, 32, seed=seed)
    return {"qweight": qw, "scales": s, "qzeros": qz}


def feed(layer, items):
    for name, arr, shard in items:
        p = getattr(layer, name)
        if shard is None:
            p.weight_loader(p, torch.from_numpy(arr))
        else:
            p.weight_loader(p, torch.from_numpy(arr), shard)


@pytest.mark.parametrize("world", [1, 2, 4])
def test_weight_loaders_match_reference_shards(meta, arrays, monkeypatch, world):
    g32 = AWQConfig(4, 32, True).get_linear_method()
    seeds = meta["loader_checkpoint_seeds"]
    for rank in range(world):
        monkeypatch.setattr(L, "get_tensor_model_parallel_world_size", lambda: world)
        monkeypatch.setattr(L, "get_tensor_model_parallel_rank", lambda: rank)
        tag = f"tp{world}r{rank}"

        def check(label, layer):
            for name in ("qweight", "scales", "qzeros"):
                want = arrays[f"loader/{label}/{tag}/{name}"]
                got = getattr(layer, name).data.numpy()
                assert got.shape == want.shape and (got.view(np.uint8) == want.view(np.uint8)).all(), \
                    (label, tag, name)

        ck = ckpt(64, 128, seeds["column"])
        layer = L.ColumnParallelLinear(64, 128, bias=False, params_dtype=torch.float16, linear_method=g32)
        feed(layer, [(n, ck[n], None) for n in ck])
        check("column", layer)

        ck = ckpt(128, 64, seeds["row"])
        layer = L.RowParallelLinear(128, 64, bias=False, params_dtype=torch.float16, linear_method=g32)
        feed(layer, [(n, ck[n], None) for n in ck])
        check("row", layer)

        a, b = ckpt(64, 64, seeds["merged"][0]), ckpt(64, 128, seeds["merged"][1])
        layer = L.MergedColumnParallelLinear(64, [64, 128], bias=False, params_dtype=torch.float16,
                                             linear_method=g32)
        feed(layer, [(n, a[n], 0) for n in a] + [(n, b[n], 1) for n in b])
        check("merged_split", layer)
        layer = L.MergedColumnParallelLinear(64, [64, 128], bias=False, params_dtype=torch.float16,
                                             linear_method=g32)
        feed(layer, [(n, np.concatenate([a[n], b[n]], axis=1), None) for n in a])
        check("merged_fused", layer)

        q, k, v = (ckpt(64, 64, seeds["qkv"][0]), ckpt(64, 32, seeds["qkv"][1]), ckpt(64, 32, seeds["qkv"][2]))
        layer = L.QKVParallelLinear(64, 16, 4, 2, bias=False, params_dtype=torch.float16, linear_method=g32)
        feed(layer, [(n, q[n], "q") for n in q] + [(n, k[n], "k") for n in k] + [(n, v[n], "v") for n in v])
        check("qkv_split", layer)
        layer = L.QKVParallelLinear(64, 16, 4, 2, bias=False, params_dtype=torch.float16, linear_method=g32)
        feed(layer, [(n, np.concatenate([q[n], k[n], v[n]], axis=1), None) for n in q])
        check("qkv_fused", layer)
        want = meta["loaders"][tag]
        assert layer.num_heads == want["qkv_num_heads"]
        assert layer.num_kv_heads == want["qkv_num_kv_heads"]
        assert layer.num_kv_head_replicas == want["qkv_num_kv_head_replicas"]
        assert layer.output_size == want["qkv_output_size"]


def test_layer_conventions(monkeypatch):
    """forward returns (output, output_bias); skip_bias_add hands the bias back; TP misalignment raises."""
    fake = FakeOps()
    monkeypatch.setattr(awq_mod, "ops", fake)
    method = AWQConfig(4, 32, True).get_linear_method()
    ck = ckpt(64, 128, 7)
    layer = L.ReplicatedLinear(64, 128, bias=True, skip_bias_add=True, params_dtype=torch.float16,
                               linear_method=method)
    for n in ck:
        getattr(layer, n).data.copy_(torch.from_numpy(ck[n]))
    layer.bias.data.fill_(0.5)
    x = torch.from_numpy(orc.make_inputs(3, 64, 128, 32, seed=8)[0])
    out, ob = layer(x)
    assert ob is layer.bias and out.shape == (3, 128)
    want = orc.gemm_faithful(x.numpy(), ck["qweight"], ck["scales"], ck["qzeros"])
    assert np.allclose(out.detach().numpy().astype(np.float32), want.astype(np.float32), rtol=1e-2, atol=1e-3)
    layer2 = L.ReplicatedLinear(64, 128, bias=True, params_dtype=torch.float16, linear_method=method)
    for n in ck:
        getattr(layer2, n).data.copy_(torch.from_numpy(ck[n]))
    layer2.bias.data.fill_(0.5)
    out2, ob2 = layer2(x)
    assert ob2 is None
    assert np.allclose(out2.detach().numpy().astype(np.float32), want.astype(np.float32) + 0.5, rtol=1e-2, atol=2e-3)
    # Llama-7B down_proj (11008 = 86 groups of 128) cannot be row-split 4 ways (awq.py:84-88)
    monkeypatch.setattr(L, "get_tensor_model_parallel_world_size", lambda: 4)
    m128 = AWQConfig(4, 128, True).get_linear_method()
    with pytest.raises(ValueError, match="input size is not aligned"):
        L.RowParallelLinear(11008, 4096, bias=False, params_dtype=torch.float16, linear_method=m128)
    with pytest.raises(ValueError, match="adding bias"):
        L.RowParallelLinear(1024, 64, bias=True, reduce_results=False, params_dtype=torch.float16,
                            linear_method=m128)
    with pytest.raises(AssertionError):
        L.set_weight_attrs(layer.qweight, {"input_dim": 1})


def test_weight_loader_after_a_forward_invalidates_the_packed_copy(monkeypatch):
    """Loaders write through `param.data`, which leaves the tensor's write counter alone; the layers'
    loaders mark the parameter as written so the W4A16 method re-packs on the next call."""
    fake = FakeOps()
    monkeypatch.setattr(awq_mod, "ops", fake)
    method = AWQConfig(4, 32, True).get_linear_method()
    layer = L.ColumnParallelLinear(64, 128, bias=False, params_dtype=torch.float16, linear_method=method)
    first, second = ckpt(64, 128, 11), ckpt(64, 128, 12)
    x = torch.from_numpy(orc.make_inputs(2, 64, 128, 32, seed=13)[0])
    for ck in (first, second):
        for n in ck:
            p = getattr(layer, n)
            p.weight_loader(p, torch.from_numpy(ck[n]))
        out, _ = layer(x)
        want = orc.gemm_faithful(x.numpy(), ck["qweight"], ck["scales"], ck["qzeros"])
        assert np.allclose(out.detach().numpy().astype(np.float32), want.astype(np.float32), rtol=1e-2, atol=1e-3)
    assert [t[0] for t in fake.trace].count("repack") == 2


def test_gate_up_silu_and_mul_uses_the_fused_call(monkeypatch):
    """A merged gate_up projection + SiluAndMul (llama.py:74-80): one fused operator call below the
    large-batch switch point, GEMM then activation above it; same values either way."""
    from vllm_quick_amd.layers.activation import SiluAndMul, gate_up_silu_and_mul
    fake = FakeOps()
    monkeypatch.setattr(awq_mod, "ops", fake)
    method = awq_mod.AWQLinearMethod(AWQConfig(4, 32, True), fp16_matmul_threshold=4)
    layer = L.MergedColumnParallelLinear(64, [64, 64], bias=False, params_dtype=torch.float16, linear_method=method)
    gate, up = ckpt(64, 64, 21), ckpt(64, 64, 22)
    for i, ck in enumerate((gate, up)):
        for n in ck:
            p = getattr(layer, n)
            p.weight_loader(p, torch.from_numpy(ck[n]), i)
    for tokens, want_ops in ((3, ["awq_gemm_silu_and_mul"]), (4, ["awq_dequantize"])):
        x = torch.from_numpy(orc.make_inputs(tokens, 64, 64, 32, seed=23)[0])
        del fake.trace[:]
        y = gate_up_silu_and_mul(layer, x)
        assert [t[0] for t in fake.trace if t[0] != "repack"] == want_ops
        g = torch.from_numpy(orc.gemm_faithful(x.numpy(), gate["qweight"], gate["scales"], gate["qzeros"]))
        u = torch.from_numpy(orc.gemm_faithful(x.numpy(), up["qweight"], up["scales"], up["qzeros"]))
        want = SiluAndMul()(torch.cat([g, u], -1))
        assert y.shape == (tokens, 64)
        assert np.allclose(y.float().numpy(), want.float().numpy(), rtol=2e-2, atol=2e-3)


def test_row_parallel_add_uses_the_fused_call(monkeypatch):
    """`residual + o_proj(x)` / `residual + down_proj(x)` of a decoder layer (llama.py:197-223): one
    fused operator call below the large-batch switch point, GEMM then add above it; the same bits as
    the two separate steps either way, and the residual is updated in place."""
    fake = FakeOps()
    monkeypatch.setattr(awq_mod, "ops", fake)
    method = awq_mod.AWQLinearMethod(AWQConfig(4, 32, True), fp16_matmul_threshold=4)
    layer = L.RowParallelLinear(64, 96, bias=True, params_dtype=torch.float16, linear_method=method)
    ck = ckpt(64, 96, 31)
    for n in ck:
        p = getattr(layer, n)
        p.weight_loader(p, torch.from_numpy(ck[n]))
    layer.bias.data.copy_(torch.linspace(-1, 1, 96).half())
    for tokens, want_ops in ((3, ["awq_gemm_add_"]), (4, ["awq_dequantize"])):
        x = torch.from_numpy(orc.make_inputs(tokens, 64, 96, 32, seed=33)[0])
        res = (torch.arange(tokens * 96).reshape(tokens, 96) % 13 - 6).half() * 0.25
        want = res + layer(x)[0]
        del fake.trace[:]
        got = L.row_parallel_add_(layer, x, res)
        assert got is res
        assert [t[0] for t in fake.trace if t[0] != "repack"] == want_ops
        assert torch.equal(got, want)
    # a layer that hands its bias back to the caller takes the two-step route
    layer.skip_bias_add = True
    x = torch.from_numpy(orc.make_inputs(2, 64, 96, 32, seed=34)[0])
    res = torch.zeros(2, 96, dtype=torch.float16)
    del fake.trace[:]
    L.row_parallel_add_(layer, x, res)
    assert [t[0] for t in fake.trace if t[0] != "repack"] == ["awq_gemm"]
    assert torch.equal(res, layer(x)[0])
