"""GPU parity of every benchmark configuration AT ITS OWN SIZE, through the C ABI of libw4a16.so,
against the CPU oracle:

* the kernel sweep on the 4096 x 11008 layer at 1 / 8 / 16 / 32 / 128 / 512 rows;
* the four linear layers of a Llama-2-7B decoder layer (vllm/model_executor/models/llama.py:63-70,
  119-132 of the reference) at 1 / 16 / 64 rows, the merged gate_up layer also through the fused
  silu-and-mul epilogue;
* the same four layers of Llama-2-70B as ONE OF 8 tensor-parallel ranks holds them;
* the column and row shards of the 4096 x 11008 layer that the multi-GPU benchmark launches;
* a Mixtral-8x7B expert block (mixtral_quant.py:71-82: 8 experts, 4096 -> 14336 twice -> 4096) as two
  batched launches, with and without the per-expert switch;
* one decoder layer of the decode harness (its four GEMMs chained).

Tolerance: rtol 1e-2 / atol 1e-3 against the reference-faithful oracle.  The faithful oracle rounds
eight split-K slices to fp16 before summing them (gemm_kernels.cu:262-270, :445), the kernels here round
once, so on long K a few outputs can differ by more than that bound through the oracle's OWN rounding.
That is bounded per element, not by a blanket share: an output outside rtol/atol must be within
`slack` of the faithful value, where `slack` is the sum of the half-ulps of the oracle's eight fp16
slices and of its final rounding plus the kernel's own single half-ulp, and must be inside rtol/atol
of the float64 result.
"""
import numpy as np
import pytest
import torch

from _parity import ATOL, RTOL, assert_parity, reference_arithmetic
from oracle import awq_oracle as orc

pytestmark = pytest.mark.gpu


def dev_t(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


@pytest.fixture(scope="module")
def ops():
    from vllm_quick_amd._C import ops
    return ops


class Layer:
    """A synthetic AWQ layer: stock tensors on the host and the device, dequantised oracle weights, packed form."""

    def __init__(self, ops, cuda, K, N, seed, G=128):
        self.K, self.N, self.G = K, N, G
        _, self.qw, self.s, self.qz = orc.make_inputs(1, K, N, G, seed=seed)
        self.w = orc.dequantize(self.qw, self.s, self.qz)
        self.t = [dev_t(a, cuda) for a in (self.qw, self.s, self.qz)]
        self.packed = ops.awq_repack(*self.t)

    def x(self, M, seed):
        return orc.make_inputs(M, self.K, 8, 8, seed=seed)[0]


_layers = {}


def layer(ops, cuda, K, N, seed=0):
    key = (K, N, seed)
    if key not in _layers:
        if len(_layers) >= 2:  # the large ones hold ~1 GB of host arrays each
            _layers.clear()
        _layers[key] = Layer(ops, cuda, K, N, seed)
    return _layers[key]


# --------------------------------------------------------------------------
# config 2: kernel sweep on the 4096 x 11008 layer
# --------------------------------------------------------------------------
@pytest.mark.parametrize("M", [1, 8, 16, 32, 128, 512])
def test_kernel_sweep_full_size(ops, cuda, M):
    L = layer(ops, cuda, 4096, 11008)
    x = L.x(M, 100 + M)
    y = ops.awq_gemm_packed(dev_t(x, cuda), L.packed).cpu().numpy()
    n = assert_parity(y, *reference_arithmetic(x, L.w), what=f"4096x11008 M={M}")
    print(f"M={M}: {n} of {y.size} outputs inside the oracle's slack only")


# --------------------------------------------------------------------------
# config 3: Llama-2-7B decoder-layer shapes; config 4: Llama-2-70B as one of 8 ranks
# --------------------------------------------------------------------------
LLAMA_7B = [("qkv_proj", 4096, 12288), ("o_proj", 4096, 4096), ("gate_up_proj", 4096, 22016),
            ("down_proj", 11008, 4096)]
LLAMA_70B_RANK = [("qkv_proj", 8192, 1280), ("o_proj", 1024, 8192), ("gate_up_proj", 8192, 7168),
                  ("down_proj", 3584, 8192)]


def silu_mul_oracle(y16):
    """silu_and_mul of the reference on an fp16 [M, 2I] GEMM result (csrc/activation_kernels.cu:9-34)."""
    half = y16.shape[1] // 2
    g, u = torch.from_numpy(y16[:, :half].copy()), torch.from_numpy(y16[:, half:].copy())
    gf = g.float()
    return ((gf / (1.0 + torch.exp(-gf))).half() * u).numpy()


def act_bound(g, u, dg, du):
    """|d silu(g) u| for |dg|, |du| perturbations of the two GEMM results (|silu'| <= 1.1), plus fp16 ulps."""
    sg = g / (1.0 + np.exp(-g))
    return 1.1 * np.abs(u) * dg + np.abs(sg) * du + 2.0**-9 * np.abs(sg * u) + 1e-6


@pytest.mark.parametrize("M", [1, 16, 64])
@pytest.mark.parametrize("name,K,N", LLAMA_7B + LLAMA_70B_RANK, ids=lambda v: str(v))
def test_llama_layer_shapes(ops, cuda, name, K, N, M):
    L = layer(ops, cuda, K, N, seed=K + N)
    x = L.x(M, 7 * M + N)
    tx = dev_t(x, cuda)
    faithful, exact, slack = reference_arithmetic(x, L.w)
    y = ops.awq_gemm_packed(tx, L.packed).cpu().numpy()
    assert_parity(y, faithful, exact, slack, what=f"{name} {K}x{N} M={M}")
    if name == "gate_up_proj":
        # merged [gate | up] weight with the activation in the GEMM epilogue (llama.py:74-80)
        a = ops.awq_gemm_packed_silu_and_mul(tx, L.packed).cpu().numpy().astype(np.float64)
        want = silu_mul_oracle(faithful).astype(np.float64)
        half = N // 2
        g, u = faithful[:, :half].astype(np.float64), faithful[:, half:].astype(np.float64)
        tol = np.maximum(ATOL + RTOL * np.abs(faithful.astype(np.float64)), slack)
        assert (np.abs(a - want) <= act_bound(g, u, tol[:, :half], tol[:, half:])).all(), f"{name} fused activation"
        # and bit-identical to this library's own two-step route on the same launch
        two_step = silu_mul_oracle(y)
        assert np.abs(a - two_step.astype(np.float64)).max() <= 2.0**-9 * max(1.0, np.abs(two_step).max())


# --------------------------------------------------------------------------
# config 4, synthetic: the shards of the 4096 x 11008 layer the multi-GPU benchmark launches
# --------------------------------------------------------------------------
@pytest.mark.parametrize("p", [2, 4, 8])
def test_benchmark_layer_shards(ops, cuda, p):
    for (K, N) in ((4096, 11008 // p), (4096 // p, 11008)):  # column shard, row shard
        L = layer(ops, cuda, K, N, seed=p)
        for M in (1, 16, 32, 128):
            x = L.x(M, p + M)
            # through the reference's operator signature (stock tensors, cached repack)
            y = ops.awq_gemm(dev_t(x, cuda), *L.t, 8).cpu().numpy()
            assert_parity(y, *reference_arithmetic(x, L.w), what=f"p={p} {K}x{N} M={M}")


# --------------------------------------------------------------------------
# config 5: Mixtral-8x7B expert block at model size, two batched launches
# --------------------------------------------------------------------------
@pytest.mark.parametrize("M", [1, 16])
def test_mixtral_experts_full_size(ops, cuda, M):
    E, H, I, G = 8, 4096, 14336, 128
    x = orc.make_inputs(M, H, 8, 8, seed=900 + M)[0]
    tx = dev_t(x, cuda)
    w13, w2 = [], []
    w13_dev, w2_dev = [], []
    for e in range(E):
        # [w1 | w3] of expert e as one [H, 2I] problem, w2 as [I, H]
        _, qw, s, qz = orc.make_inputs(1, H, 2 * I, G, seed=1000 + e)
        w13.append(orc.dequantize(qw, s, qz))
        w13_dev.append([dev_t(a, cuda) for a in (qw, s, qz)])
        _, qw, s, qz = orc.make_inputs(1, I, H, G, seed=2000 + e)
        s = (s.astype(np.float32) * 0.25).astype(np.float16)  # keeps |y| = O(1) over the long K
        w2.append(orc.dequantize(qw, s, qz))
        w2_dev.append([dev_t(a, cuda) for a in (qw, s, qz)])
    b13 = ops.awq_repack_batch(*[[t[i] for t in w13_dev] for i in range(3)])
    b2 = ops.awq_repack_batch(*[[t[i] for t in w2_dev] for i in range(3)])
    del w13_dev, w2_dev
    act = ops.awq_gemm_packed_silu_and_mul(tx, b13)  # [E, M, I]
    y = ops.awq_gemm_packed_batched(act, b2)  # [E, M, H]
    act_h, y_h = act.cpu().numpy(), y.cpu().numpy()
    for e in range(E):
        faithful, exact, slack = reference_arithmetic(x, w13[e])
        want = silu_mul_oracle(faithful).astype(np.float64)
        g, u = faithful[:, :I].astype(np.float64), faithful[:, I:].astype(np.float64)
        tol = np.maximum(ATOL + RTOL * np.abs(faithful.astype(np.float64)), slack)
        assert (np.abs(act_h[e].astype(np.float64) - want) <= act_bound(g, u, tol[:, :I], tol[:, I:])).all(), e
        # second launch: checked on the activations the first launch produced
        assert_parity(y_h[e], *reference_arithmetic(act_h[e], w2[e]), what=f"expert {e} w2 M={M}")
    # the per-expert switch: skipped experts are never read and count as zero; with the launch shape of the
    # unmasked call (no hint) the others are bit-identical, with the shape picked for 3 busy experts they
    # meet the same bound against the oracle
    mask_h = [1, 0, 0, 1, 0, 1, 0, 0]
    mask = torch.tensor(mask_h, dtype=torch.int32, device=cuda)
    for hint in (0, 3):
        act_m = ops.awq_gemm_packed_silu_and_mul(tx, b13, mask=mask, active_hint=hint)
        y_m = ops.awq_gemm_packed_batched(act_m, b2, mask=mask, active_hint=hint)
        for e in range(E):
            if not mask_h[e]:
                assert not act_m[e].any() and not y_m[e].any(), e
            elif hint == 0:
                assert torch.equal(act_m[e], act[e]) and torch.equal(y_m[e], y[e]), e
            else:
                assert_parity(y_m[e].cpu().numpy(), *reference_arithmetic(act_m[e].cpu().numpy(), w2[e]),
                              what=f"masked expert {e} w2 M={M}")


# --------------------------------------------------------------------------
# config 3, harness: the four GEMMs of one decoder layer of the decode benchmark, chained
# --------------------------------------------------------------------------
@pytest.mark.parametrize("bs", [1, 8])
def test_decode_harness_layer_linears(ops, cuda, bs):
    from benchmarks.llama_decode import SHAPES, DecoderLayer, fill_awq
    from vllm_quick_amd.layers.quantization import AWQConfig
    cfg = SHAPES["7b"]
    method = AWQConfig(4, 128, True).get_linear_method()
    lay = DecoderLayer(cfg, method, 8, 16, cuda).to(cuda)
    gen = torch.Generator(device=cuda)
    gen.manual_seed(5)
    lins = (lay.qkv_proj, lay.o_proj, lay.gate_up_proj, lay.down_proj)
    # the harness's scales (a tenth of the usual range) are raised per layer so that every GEMM of the chain
    # sees and produces values of order 1, where the absolute part of the tolerance does not trivialise the check
    for lin, up in zip(lins, (10.0, 2.5, 2.5, 3.0)):
        fill_awq(lin, gen, cuda)
        lin.scales.data = (lin.scales.data.float() * up).half()
    x = orc.make_inputs(bs, cfg["hidden"], 8, 8, seed=bs)[0]
    got = lay.linears_only(dev_t(x, cuda)).cpu().numpy()
    # the same chain on the oracle; every GEMM is checked on the inputs the device chain fed it, so that
    # the bound stays the single-GEMM one
    w = [orc.dequantize(*[getattr(lin, n).detach().cpu().numpy() for n in ("qweight", "scales", "qzeros")])
         for lin in lins]
    a = lay.qkv_proj(dev_t(x, cuda))[0]
    assert_parity(a.cpu().numpy(), *reference_arithmetic(x, w[0]), what="qkv")
    a_in = a[:, :cfg["hidden"]].contiguous()
    b = lay.o_proj(a_in)[0]
    assert_parity(b.cpu().numpy(), *reference_arithmetic(a_in.cpu().numpy(), w[1]), what="o_proj")
    gu = lay.gate_up_proj(b)[0]
    assert_parity(gu.cpu().numpy(), *reference_arithmetic(b.cpu().numpy(), w[2]), what="gate_up")
    from vllm_quick_amd.layers.activation import gate_up_silu_and_mul
    c = gate_up_silu_and_mul(lay.gate_up_proj, b)
    want_c = silu_mul_oracle(gu.cpu().numpy())
    assert np.abs(c.cpu().numpy().astype(np.float64) - want_c).max() <= 2.0**-9 * max(1.0, np.abs(want_c).max())
    d = lay.down_proj(c)[0]
    assert_parity(d.cpu().numpy(), *reference_arithmetic(c.cpu().numpy(), w[3]), what="down_proj")
    assert torch.equal(dev_t(got, cuda), d)  # linears_only is exactly this chain
