"""GPU parity of the add-to-out epilogue (W4A16_EPI_ADD_TO_OUT of w4a16_gemm_packed_f16_ex, called
through the C ABI): `residual + linear(x)` of a decoder layer in one launch.

Two checks.  Bit-exact: the fused launch must equal residual + (the same launch shape without the
epilogue), added by torch in fp16, because the epilogue rounds the GEMM result to fp16, adds the bias in
fp16 and then adds the residual in fp16, which is what the two separate operations do.  Against the
reference-faithful oracle: the GEMM tolerance (rtol 1e-2 / atol 1e-3 on the GEMM result) plus one fp16
ulp of the sum."""
import numpy as np
import pytest
import torch

from oracle import awq_oracle as orc

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-2, 1e-3


@pytest.fixture(scope="module")
def L():
    from vllm_quick_amd import _lib
    return _lib


@pytest.fixture(scope="module")
def ops():
    from vllm_quick_amd._C import ops
    return ops


def run_ex(L, x, pk, out, bias, epilogue, cfg):
    M = x.shape[0]
    rc = L.lib.w4a16_gemm_packed_f16_ex(x.data_ptr(), 0, pk.packed_w.data_ptr(), pk.packed_sz.data_ptr(),
                                        None if bias is None else bias.data_ptr(), out.data_ptr(), M * pk.N, M, pk.K,
                                        pk.N, pk.G, 1, epilogue, *cfg, torch.cuda.current_stream().cuda_stream)
    L.check(rc, "gemm_ex")
    return out


CASES = [  # M, K, N, G, bias?, launch shapes to force (None = the library's choice)
    (1, 512, 256, 128, False, [None, (1, 1, 16, 0), (1, 3, 8, 0), (1, 2, 4, 0)]),
    (5, 384, 440, 128, True, [None, (1, 6, 8, 0), (1, 4, 4, 0)]),
    (20, 256, 200, 64, False, [None, (2, 1, 16, 0), (2, 4, 8, 0)]),
    (33, 11008, 96, 128, True, [None, (2, 3, 1, 8)]),
    (70, 640, 416, 32, True, [None, (4, 3, 2, 4), (4, 3, 4, 2), (2, 3, 2, 4), (2, 3, 8, 0), (2, 3, 1, -8), (1, 2, 1, -8),
                              (2, 2, 1, -4), (4, 1, 8, -1), (2, 3, 8, -1)]),
    (130, 1024, 1376, 128, False, [None]),
    (300, 512, 512, 128, False, [None]),
]


@pytest.mark.parametrize("M,K,N,G,with_bias,cfgs", CASES)
def test_add_to_out_through_c_abi(L, ops, cuda, M, K, N, G, with_bias, cfgs):
    x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=M + N)
    rng = np.random.default_rng(M)
    res = (rng.standard_normal((M, N)) * 2).astype(np.float16)
    bias = (rng.standard_normal(N) * 0.5).astype(np.float16) if with_bias else None
    tx, tres = torch.from_numpy(x).to(cuda), torch.from_numpy(res).to(cuda)
    tb = None if bias is None else torch.from_numpy(bias).to(cuda)
    pk = ops.awq_repack(*[torch.from_numpy(a).to(cuda) for a in (qw, s, qz)])
    gemm = orc.gemm_faithful(x, qw, s, qz).astype(np.float32)
    if bias is not None:
        gemm = (torch.from_numpy(gemm).half() + torch.from_numpy(bias)).float().numpy()
    for cfg in cfgs:
        c = (0, 0, 0, 0) if cfg is None else cfg
        plain = run_ex(L, tx, pk, torch.empty((M, N), dtype=torch.float16, device=cuda), tb, 0, c)
        fused = run_ex(L, tx, pk, tres.clone(), tb, 2, c)
        torch.cuda.synchronize()
        assert torch.equal(fused, tres + plain), f"cfg {cfg}: not the fp16 sum of residual and GEMM result"
        got = fused.float().cpu().numpy()
        err = np.abs(got - (res.astype(np.float32) + gemm))
        lim = ATOL + RTOL * np.abs(gemm) + 2.0**-10 * np.abs(got) + 1e-6
        assert (err <= lim).all(), f"cfg {cfg}: max excess {(err - lim).max():.3e}"


def test_operator_and_method_update_the_residual_in_place(ops, cuda):
    from vllm_quick_amd.layers import linear as Lin
    from vllm_quick_amd.layers.quantization import AWQConfig
    M, K, N, G = 6, 256, 384, 128
    x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=9)
    tx = torch.from_numpy(x).to(cuda)
    pk = ops.awq_repack(*[torch.from_numpy(a).to(cuda) for a in (qw, s, qz)])
    res = torch.ones((M, N), dtype=torch.float16, device=cuda)
    want = res + ops.awq_gemm_packed(tx, pk)
    r = ops.awq_gemm_packed_add_(res, tx, pk)
    assert r is res and torch.equal(res, want)
    with pytest.raises(ValueError):
        ops.awq_gemm_packed_add_(res[:, :N - 8], tx, pk)
    # through the layer helper, [batch, seq, hidden] activations
    layer = Lin.RowParallelLinear(K, N, bias=True, params_dtype=torch.float16,
                                  linear_method=AWQConfig(4, G, True).get_linear_method()).to(cuda)
    for name, arr in (("qweight", qw), ("scales", s), ("qzeros", qz)):
        p = getattr(layer, name)
        p.weight_loader(p, torch.from_numpy(arr))
    layer.bias.data.copy_(torch.linspace(-1, 1, N).half())
    x3 = tx.view(2, 3, K)
    res3 = torch.full((2, 3, N), 0.5, dtype=torch.float16, device=cuda)
    want3 = res3 + layer(x3)[0]
    got3 = Lin.row_parallel_add_(layer, x3, res3)
    assert got3 is res3 and torch.equal(res3, want3)
