"""bf16 activations (w4a16_gemm_packed_bf16 of the C ABI, `ops.awq_gemm` / `ops.awq_gemm_packed` with bf16 inputs).

PARITY UNPINNED by construction: the reference's AWQ path takes fp16 only
(vllm/model_executor/layers/quantization/awq.py:42-43), so nothing of the reference pins these results.  The tests
check the extension against its own definition, restated in oracle/awq_oracle.py:
  * weights  w = bf16((q - z) * s), one rounding from the exact product: BIT-EXACT, read back through unit rows
    (x = e_k makes output row k the weight row k, a single exact product per output);
  * GEMM     float64 sum of w * x; the kernel accumulates in fp32 and rounds to bf16 once, so an output may differ
    from the float64 value by half a bf16 ulp plus the fp32 accumulation error, bounded here by
    K * 2^-23 * sum |w x| (every partial sum rounded to 24 bits, far above what occurs);
  * add-to-out and bias: bf16 + bf16 rounded once, checked the same way on top of the rounded GEMM result.
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import awq_oracle as orc

pytestmark = pytest.mark.gpu


def dev_t(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


@pytest.fixture(scope="module")
def L():
    from vllm_quick_amd import _lib
    return _lib


@pytest.fixture(scope="module")
def ops():
    from vllm_quick_amd._C import ops
    return ops


def built_shapes(L):
    out, buf, i = [], (ctypes.c_int * 3)(), 0
    while L.lib.w4a16_built_shape_bf16(i, buf) == 0:
        out.append(tuple(buf))
        i += 1
    return out


DECODE_TAG = -16  # first entry of a built shape that names the decode kernel (up to 16 rows, whole 128-k blocks)


def refused(cfg, M, K):
    """Launch shapes that do not take the problem: the 32x32x16 and decode kernels want whole 128-k blocks, the decode
    kernel up to 16 rows.  (-36, t, nw) names the shared-slab 32x32x16 kernel: whole blocks, any row count.)"""
    if cfg is None or cfg[0] >= 0:
        return False
    if K % 128 != 0:
        return True
    if cfg[0] == DECODE_TAG:  # up to 16 rows; 17-32 with one column tile x 16 waves
        return M > 32 or (M > 16 and not (cfg[2] == 16 and cfg[1] == 1))
    return False


def bf16_tensor(bits, cuda):
    return dev_t(bits.view(np.int16), cuda).view(torch.bfloat16)


def bits_of(t):
    return t.cpu().view(torch.int16).numpy().view(np.uint16)


def half_ulp_bf16(v):
    """Half a unit in the last place (8 significant bits) of |v| (float64), at least that of the smallest normal."""
    a = np.maximum(np.abs(v), 2.0**-126)
    return 2.0**(np.floor(np.log2(a)) - 8)


def make(M, K, N, G, seed, negative_scales=False):
    x16, qw, s, qz = orc.make_inputs(M, K, N, G, seed=seed)
    if negative_scales:
        s = s.copy()
        s[:, ::3] = -s[:, ::3]
    xb = orc.f32_to_bf16_bits(x16.astype(np.float32))
    return xb, qw, s, qz


def check_gemm(got_bits, xb, qw, s, qz, what, base_bits=None, bias_bits=None):
    exact = orc.gemm_bf16_exact(xb, qw, s, qz)
    mag = orc.gemm_bf16_abs(xb, qw, s, qz)
    K = xb.shape[1]
    got = orc.bf16_bits_to_f32(got_bits).astype(np.float64)
    bound = half_ulp_bf16(exact) * 1.001 + K * 2.0**-23 * mag
    want = exact
    if bias_bits is not None:  # bf16(bf16(sum) + bias): two roundings
        b = orc.bf16_bits_to_f32(bias_bits).astype(np.float64)[None, :]
        want = exact + b
        bound = bound + half_ulp_bf16(want) * 1.001
    if base_bits is not None:  # out + result: one more rounding
        base = orc.bf16_bits_to_f32(base_bits).astype(np.float64)
        want = want + base
        bound = bound + half_ulp_bf16(want) * 1.001
    d = np.abs(got - want)
    assert (d <= bound).all(), f"{what}: {int((d > bound).sum())} of {d.size} outputs beyond the rounding bound " \
                               f"(worst {float((d / bound).max()):.2f}x)"


@pytest.mark.parametrize("G", [128, 64, 32])
def test_weights_bit_exact_through_unit_rows(L, ops, cuda, G):
    """x = e_k for 64 values of k spread over K: the output rows are weight rows, bit for bit; every built launch
    shape, positive and negative scales."""
    K, N = 1024, 400
    _, qw, s, qz = make(1, K, N, G, seed=5, negative_scales=True)
    want = orc.dequantize_bf16(qw, s, qz)
    canon = np.where(want == 0x8000, 0, want).astype(np.uint16)  # (a sum that starts from +0 cannot return -0)
    packed = ops.awq_repack(dev_t(qw, cuda), dev_t(s, cuda), dev_t(qz, cuda))
    ks = np.unique(np.concatenate([np.arange(0, K, 37), [K - 1, 127, 128, 31, 32]]))[:64]
    for M in (1, 16, len(ks)):
        rows = ks[:M]
        xb = np.zeros((M, K), np.uint16)
        xb[np.arange(M), rows] = 0x3F80  # 1.0
        tx = bf16_tensor(xb, cuda)
        for cfg in [None] + built_shapes(L):
            if refused(cfg, M, K):
                continue
            y = ops.awq_gemm_packed_bf16(tx, packed, cfg=cfg)
            assert y.dtype == torch.bfloat16 and tuple(y.shape) == (M, N)
            got = bits_of(y)
            bad = got != canon[rows]
            assert not bad.any(), f"G={G} M={M} cfg={cfg}: {int(bad.sum())} differ, e.g. got {got[bad][:4]} want {canon[rows][bad][:4]}"


@pytest.mark.parametrize("M", [1, 3, 16, 17, 33, 128, 300, 1100])
def test_gemm_within_rounding_of_float64(L, ops, cuda, M):
    K, N, G = 4096, 1376, 128
    xb, qw, s, qz = make(M, K, N, G, seed=40 + M)
    packed = ops.awq_repack(dev_t(qw, cuda), dev_t(s, cuda), dev_t(qz, cuda))
    y = ops.awq_gemm_packed_bf16(bf16_tensor(xb, cuda), packed)
    check_gemm(bits_of(y), xb, qw, s, qz, f"M={M}")
    y2 = ops.awq_gemm_packed_bf16(bf16_tensor(xb, cuda), packed)
    assert torch.equal(y.view(torch.int16), y2.view(torch.int16)), "same launch, same bits"


def test_every_built_shape_and_odd_sizes(L, ops, cuda):
    """K not a multiple of 128 (streaming shapes; the 32x32x16 and decode shapes, listed with a negative first entry,
    take whole 128-k blocks and refuse it, the decode shapes also more than 16 rows) and K = 1280, N not a multiple of
    16 per workgroup, every built shape at 5, 16, 40 and 150 rows."""
    N, G = 328, 32
    for K in (1056, 1280):
        for M in (5, 16, 27, 40, 150):
            xb, qw, s, qz = make(M, K, N, G, seed=7 + M + K)
            packed = ops.awq_repack(dev_t(qw, cuda), dev_t(s, cuda), dev_t(qz, cuda))
            for cfg in built_shapes(L):
                if refused(cfg, M, K):
                    with pytest.raises(ValueError):
                        ops.awq_gemm_packed_bf16(bf16_tensor(xb, cuda), packed, cfg=cfg)
                    continue
                y = ops.awq_gemm_packed_bf16(bf16_tensor(xb, cuda), packed, cfg=cfg)
                check_gemm(bits_of(y), xb, qw, s, qz, f"M={M} K={K} cfg={cfg}")
            y = ops.awq_gemm_packed_bf16(bf16_tensor(xb, cuda), packed)  # the library's choice
            check_gemm(bits_of(y), xb, qw, s, qz, f"M={M} K={K} default")


def test_bias_and_add_to_out(L, ops, cuda):
    M, K, N, G = 9, 2048, 512, 128
    xb, qw, s, qz = make(M, K, N, G, seed=3)
    rng = np.random.default_rng(9)
    bias = orc.f32_to_bf16_bits(rng.standard_normal(N).astype(np.float32))
    base = orc.f32_to_bf16_bits(rng.standard_normal((M, N)).astype(np.float32))
    packed = ops.awq_repack(dev_t(qw, cuda), dev_t(s, cuda), dev_t(qz, cuda))
    tx = bf16_tensor(xb, cuda)
    y = ops.awq_gemm_packed_bf16(tx, packed, bias=bf16_tensor(bias, cuda))
    check_gemm(bits_of(y), xb, qw, s, qz, "bias", bias_bits=bias)
    out = bf16_tensor(base, cuda).clone()
    r = ops.awq_gemm_packed_bf16(tx, packed, out=out)
    assert r.data_ptr() == out.data_ptr()
    check_gemm(bits_of(out), xb, qw, s, qz, "add to out", base_bits=base)
    # add-to-out is exactly the two separate bf16 steps
    plain = ops.awq_gemm_packed_bf16(tx, packed)
    assert torch.equal((bf16_tensor(base, cuda) + plain).view(torch.int16), out.view(torch.int16))


def test_operator_mirror_and_errors(L, ops, cuda):
    M, K, N, G = 4, 512, 256, 128
    xb, qw, s, qz = make(M, K, N, G, seed=12)
    tqw, ts, tqz = dev_t(qw, cuda), dev_t(s, cuda), dev_t(qz, cuda)
    tx = bf16_tensor(xb, cuda)
    y = ops.awq_gemm(tx, tqw, ts, tqz, 8)  # the reference's operator signature, bf16 activations
    assert y.dtype == torch.bfloat16
    check_gemm(bits_of(y), xb, qw, s, qz, "ops.awq_gemm")
    packed = ops.awq_repack(tqw, ts, tqz)
    assert torch.equal(ops.awq_gemm_packed(tx, packed).view(torch.int16), y.view(torch.int16))
    with pytest.raises(RuntimeError):  # a bias of the other type
        ops.awq_gemm_packed_bf16(tx, packed, bias=torch.zeros(N, dtype=torch.float16, device=cuda))
    with pytest.raises(RuntimeError):  # fp16 activations through the bf16 entry
        ops.awq_gemm_packed_bf16(tx.to(torch.float16), packed)
    with pytest.raises(ValueError):  # a launch shape that is not built
        ops.awq_gemm_packed_bf16(tx, packed, cfg=(1, 5, 16))
    # the fused activation is not built for bf16: refused through the C ABI
    out = torch.empty((M, N // 2), dtype=torch.bfloat16, device=cuda)
    rc = L.lib.w4a16_gemm_packed_bf16(tx.data_ptr(), packed.packed_w.data_ptr(), packed.packed_sz.data_ptr(), None,
                                      out.data_ptr(), M, K, N, G, 1, 0, 0, 0, None)
    assert rc == -5


def test_full_size_layer(L, ops, cuda):
    """The benchmark layer (4096 x 11008, group 128) at 1, 16, 128 and 512 rows."""
    K, N, G = 4096, 11008, 128
    _, qw, s, qz = make(1, K, N, G, seed=77)
    packed = ops.awq_repack(dev_t(qw, cuda), dev_t(s, cuda), dev_t(qz, cuda))
    for M in (1, 16, 128, 512):
        xb = orc.f32_to_bf16_bits((np.random.default_rng(M).standard_normal((M, K)) * 0.25).astype(np.float32))
        y = ops.awq_gemm_packed_bf16(bf16_tensor(xb, cuda), packed)
        check_gemm(bits_of(y), xb, qw, s, qz, f"full size M={M}")


def test_compiled_binding_takes_bf16(ops, cuda):
    """`_C_native.ops.awq_gemm` (the compiled module with the reference's operator signature) dispatches on the
    activation type: bit-identical to the ctypes route."""
    from vllm_quick_amd import _C_native as native
    M, K, N, G = 7, 1024, 512, 128
    xb, qw, s, qz = make(M, K, N, G, seed=21)
    tq, ts, tz = dev_t(qw, cuda), dev_t(s, cuda), dev_t(qz, cuda)
    tx = bf16_tensor(xb, cuda)
    y = native.ops.awq_gemm(tx, tq, ts, tz, 8)
    assert y.dtype == torch.bfloat16 and tuple(y.shape) == (M, N)
    assert torch.equal(y.view(torch.int16), ops.awq_gemm_packed_bf16(tx, ops.awq_repack(tq, ts, tz)).view(torch.int16))
    with pytest.raises(RuntimeError):
        native.ops.awq_gemm(tx.float(), tq, ts, tz, 8)


def test_linear_method_with_bf16_activations(ops, cuda):
    """`AWQLinearMethod` on bf16 activations: apply_weights through the bf16 kernels, the gated MLP's activation as two
    separate steps (the fused epilogue is fp16 only), `residual += linear(x)` through the add-to-out epilogue."""
    from vllm_quick_amd.layers.activation import gate_up_silu_and_mul
    from vllm_quick_amd.layers.linear import MergedColumnParallelLinear, RowParallelLinear
    from vllm_quick_amd.layers.quantization import AWQConfig
    method = AWQConfig(4, 128, True).get_linear_method()
    H, I, M = 512, 768, 6
    gate_up = MergedColumnParallelLinear(H, [I, I], bias=False, params_dtype=torch.float16, linear_method=method).to(cuda)
    down = RowParallelLinear(I, H, bias=False, params_dtype=torch.float16, linear_method=method).to(cuda)
    raw = {}
    for name, lin, (K, N) in (("gu", gate_up, (H, 2 * I)), ("down", down, (I, H))):
        _, qw, s, qz = make(1, K, N, 128, seed=K)
        raw[name] = (qw, s, qz)
        lin.qweight.data.copy_(dev_t(qw, cuda))
        lin.scales.data.copy_(dev_t(s, cuda))
        lin.qzeros.data.copy_(dev_t(qz, cuda))
    xb = orc.f32_to_bf16_bits((np.random.default_rng(4).standard_normal((M, H)) * 0.5).astype(np.float32))
    tx = bf16_tensor(xb, cuda)
    y = gate_up(tx)[0]
    assert y.dtype == torch.bfloat16
    check_gemm(bits_of(y), xb, *raw["gu"], "gate_up")
    act = gate_up_silu_and_mul(gate_up, tx)
    assert act.dtype == torch.bfloat16
    assert torch.equal(act.view(torch.int16), (torch.nn.functional.silu(y[:, :I]) * y[:, I:]).view(torch.int16))
    base = orc.f32_to_bf16_bits(np.random.default_rng(5).standard_normal((M, H)).astype(np.float32))
    res = bf16_tensor(base, cuda).clone()
    r = method.apply_weights_add_(down.linear_weights, act, res)
    assert r.data_ptr() == res.data_ptr()
    check_gemm(bits_of(res), bits_of(act), *raw["down"], "down + residual", base_bits=base)
