"""Host logic of the mixture-of-experts block without a GPU: the native operators are replaced by
oracle-backed fakes (tests/_fakes.py), and the block's output is checked against a numpy
restatement of the reference's loop (vllm/model_executor/models/mixtral_quant.py:133-160)."""
import numpy as np
import torch

from oracle import awq_oracle as orc
from _fakes import FakeOps
from vllm_quick_amd.layers import experts as X
from vllm_quick_amd.layers.quantization import AWQConfig
from vllm_quick_amd.layers.quantization import awq as awq_mod


def fill(moe, seed):
    """Seeded AWQ tensors for every local expert and a dense router."""
    rng = np.random.default_rng(seed)
    ck = {}
    for idx in moe.expert_indicies:
        for j, name in enumerate(("w1", "w2", "w3")):
            lin = getattr(moe.experts[idx], name)
            K, N = lin.input_size, lin.output_size
            _, qw, s, qz = orc.make_inputs(1, K, N, 32, seed=seed * 100 + idx * 3 + j)
            s = (s.astype(np.float32) * 4).astype(np.float16)
            for pname, arr in (("qweight", qw), ("scales", s), ("qzeros", qz)):
                with torch.no_grad():
                    getattr(lin, pname).copy_(torch.from_numpy(arr))
            ck[(idx, name)] = (qw, s, qz)
    gate = rng.standard_normal(moe.gate.weight.shape).astype(np.float16)
    moe.gate.weight.data.copy_(torch.from_numpy(gate))
    return ck, gate


def reference_block(x, gate, ck, experts, top_k):
    """mixtral_quant.py:137-157 restated with the oracle's GEMM: fp16 tensors, fp32 routing."""
    logits = torch.from_numpy(x) @ torch.from_numpy(gate).t()
    w = torch.softmax(logits, dim=1, dtype=torch.float)
    w, sel = torch.topk(w, top_k, dim=-1)
    w /= w.sum(dim=-1, keepdim=True)
    final = None
    for idx in experts:
        ew = (w * (sel == idx)).sum(dim=-1, keepdim=True)
        h1 = torch.from_numpy(orc.gemm_faithful(x, *ck[(idx, "w1")]))
        h3 = torch.from_numpy(orc.gemm_faithful(x, *ck[(idx, "w3")]))
        act = torch.nn.functional.silu(h1) * h3
        cur = torch.from_numpy(orc.gemm_faithful(act.numpy(), *ck[(idx, "w2")])).mul_(ew)
        final = cur if final is None else final.add_(cur)
    return final.numpy()


def make(monkeypatch, E=4, top_k=2, H=64, I=96):
    fake = FakeOps()
    monkeypatch.setattr(awq_mod, "ops", fake)
    monkeypatch.setattr(X, "ops", fake)
    method = AWQConfig(4, 32, True).get_linear_method()
    moe = X.MixtralMoE(E, top_k, H, I, linear_method=method, params_dtype=torch.float16)
    return moe, fake


def test_parameter_tree_matches_the_reference_names(monkeypatch):
    moe, _ = make(monkeypatch)
    names = {n for n, _ in moe.named_parameters()}
    assert "gate.weight" in names
    for idx in range(4):
        for w in ("w1", "w2", "w3"):
            for p in ("qweight", "qzeros", "scales"):
                assert f"experts.{idx}.{w}.{p}" in names
    assert moe.experts[0].w1.qweight.shape == (64, 96 // 8)
    assert moe.experts[0].w2.qweight.shape == (96, 64 // 8)


def test_batched_block_matches_reference_loop(monkeypatch):
    moe, fake = make(monkeypatch)
    moe.sparse_max_tokens = 0  # every expert runs, as in the reference
    ck, gate = fill(moe, seed=3)
    x = (np.random.default_rng(5).standard_normal((2, 3, 64)) * 0.5).astype(np.float16)
    y = moe(torch.from_numpy(x))
    assert y.shape == (2, 3, 64) and y.dtype == torch.float16
    calls = [t for t in fake.trace if t[0].startswith("awq_gemm")]
    # two launches in all: E [w1 | w3] problems on the shared tokens with the activation fused,
    # E down projections
    assert calls == [["awq_gemm_silu_and_mul", [6, 64]], ["awq_gemm_batched", [4, 6, 96], 4]]
    want = reference_block(x.reshape(6, 64), gate, ck, moe.expert_indicies, 2)
    assert np.allclose(y.reshape(6, 64).numpy().astype(np.float32), want.astype(np.float32), rtol=1e-2, atol=2e-3)
    # the re-layout is cached until a weight is written again
    moe(torch.from_numpy(x))
    assert [t[0] for t in fake.trace].count("repack_batch") == 2
    with torch.no_grad():
        moe.experts[1].w2.qweight.add_(0)
    moe(torch.from_numpy(x))
    assert [t[0] for t in fake.trace].count("repack_batch") == 4


def test_few_tokens_skip_the_experts_nobody_selected(monkeypatch):
    """With one or two tokens the launches carry a per-expert switch: exactly the selected experts are on,
    and the block's output is what running every expert (and multiplying by zero) gives."""
    moe, fake = make(monkeypatch)
    ck, gate = fill(moe, seed=12)
    for tokens in (1, 2):
        x = (np.random.default_rng(20 + tokens).standard_normal((1, tokens, 64)) * 0.5).astype(np.float16)
        del fake.trace[:]
        y = moe(torch.from_numpy(x))
        calls = [t for t in fake.trace if t[0].startswith("awq_gemm")]
        assert [c[0] for c in calls] == ["awq_gemm_silu_and_mul", "awq_gemm_batched"]
        mask = calls[0][-1]
        assert mask == calls[1][-1] and len(mask) == 4
        logits = torch.from_numpy(x.reshape(tokens, 64)).float() @ torch.from_numpy(gate).float().t()
        chosen = set(torch.topk(logits, 2, dim=-1).indices.flatten().tolist())
        assert {i for i, on in enumerate(mask) if on} == chosen and 2 <= len(chosen) < 4
        moe.sparse_max_tokens = 0
        dense = moe(torch.from_numpy(x))
        moe.sparse_max_tokens = 4
        assert torch.equal(y, dense)
        want = reference_block(x.reshape(tokens, 64), gate, ck, moe.expert_indicies, 2)
        assert np.allclose(y.reshape(tokens, 64).numpy().astype(np.float32), want.astype(np.float32), rtol=1e-2,
                           atol=2e-3)
    # the separate-activation route switches its 2 E gate/up problems the same way
    moe.fuse_act = False
    del fake.trace[:]
    y2 = moe(torch.from_numpy(x))
    calls = [t for t in fake.trace if t[0].startswith("awq_gemm")]
    assert calls[0][-1] == mask + mask and calls[1][-1] == mask
    assert np.allclose(y2.float().numpy(), y.float().numpy(), rtol=1e-2, atol=2e-3)


def test_stamp_cache_and_invalidation(monkeypatch):
    from vllm_quick_amd.layers.linear import mark_written
    moe, fake = make(monkeypatch)
    fill(moe, seed=21)
    x = torch.from_numpy((np.random.default_rng(1).standard_normal((1, 3, 64)) * 0.5).astype(np.float16))
    count = lambda: [t[0] for t in fake.trace].count("repack_batch")  # noqa: E731
    moe(x), moe(x)
    assert count() == 2
    # a loader write that changes nothing the stamp sees but the epoch: full look, same tensors, no re-layout
    mark_written(moe.gate.weight)
    moe(x)
    assert count() == 2
    # a loader write into an expert tensor: re-layout
    lin = moe.experts[2].w1
    lin.qweight.data.copy_(lin.qweight.data)
    mark_written(lin.qweight)
    moe(x)
    assert count() == 4
    # a tensor whose storage was swapped (`param.data = ..`, what `module.to(..)` does): seen by its address
    lin.scales.data = lin.scales.data.clone()
    moe(x)
    assert count() == 6
    moe(x)
    assert count() == 6
    # a write through `.data` into the same storage touches neither address nor counter: that needs invalidate()
    lin.scales.data.copy_(lin.scales.data * 2)
    moe(x)
    assert count() == 6
    moe.invalidate()
    moe(x)
    assert count() == 8


def test_both_routes_combine_the_experts_the_same_way(monkeypatch):
    """The batched route and the per-expert loop (taken when no re-layout is possible, e.g. the first call
    under a graph capture) scale each expert's output in fp16 and sum in fp32, rounding once: the same
    values up to the order of an fp32 sum."""
    moe, fake = make(monkeypatch)
    fill(moe, seed=31)
    moe.sparse_max_tokens = 0
    x = torch.from_numpy((np.random.default_rng(2).standard_normal((1, 7, 64)) * 0.5).astype(np.float16))
    batched = moe(x)
    moe.batched = False
    looped = moe(x)
    a, b = batched.float().numpy(), looped.float().numpy()
    ulp = np.maximum(np.abs(a), 2.0 ** -14) * 2.0 ** -10
    assert (np.abs(a - b) <= ulp).all()


def test_separate_activation_route(monkeypatch):
    moe, fake = make(monkeypatch)
    assert moe.fuse_act  # K of w2 is a multiple of the group size, hence of 16: always tile-aligned
    moe.fuse_act = False
    moe.sparse_max_tokens = 0
    ck, gate = fill(moe, seed=8)
    x = (np.random.default_rng(9).standard_normal((1, 4, 64)) * 0.5).astype(np.float16)
    y = moe(torch.from_numpy(x))
    calls = [t for t in fake.trace if t[0].startswith("awq_gemm")]
    assert calls == [["awq_gemm_batched", [4, 64], 8], ["awq_gemm_batched", [4, 4, 96], 4]]
    want = reference_block(x.reshape(4, 64), gate, ck, moe.expert_indicies, 2)
    assert np.allclose(y.reshape(4, 64).numpy().astype(np.float32), want.astype(np.float32), rtol=1e-2, atol=2e-3)


def test_unbatched_fallback_is_the_reference_loop(monkeypatch):
    moe, fake = make(monkeypatch)
    ck, gate = fill(moe, seed=4)
    moe.batched = False
    x = (np.random.default_rng(6).standard_normal((1, 5, 64)) * 0.5).astype(np.float16)
    y = moe(torch.from_numpy(x))
    assert [t[0] for t in fake.trace if t[0].startswith("awq_gemm")] == ["awq_gemm"] * 12
    want = reference_block(x.reshape(5, 64), gate, ck, moe.expert_indicies, 2)
    assert np.allclose(y.reshape(5, 64).numpy().astype(np.float32), want.astype(np.float32), rtol=1e-2, atol=2e-3)
