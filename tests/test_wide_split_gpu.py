"""GPU parity of the private-slab 32x32x16 kernel with K split over workgroups
(w4a16_gemm_packed_f16_wide_split_cfg / w4a16_gemm_packed_f16_ws of the C ABI) against the oracle: every 4-wave
launch shape at 2, 3 and 4 workgroups per output tile, ragged rows and columns, a K that does not divide evenly,
bias / fused-activation / add-to-out epilogues, and the automatic choice on a few-column long-K layer.

Tolerance as for every GEMM here: rtol 1e-2 / atol 1e-3 against the reference-faithful oracle.  A split launch sums
the same fp32 products in another order than the unsplit one (two fp16 ulps apart at most), and returns the same
bits every time: the partial sums are added in split order, not in arrival order.
"""
import ctypes

import numpy as np
import pytest
import torch

from _parity import assert_parity as assert_parity_full
from _parity import reference_arithmetic
from oracle import awq_oracle as orc
from test_decode_gpu import assert_parity, dev_t
from test_fused_act_gpu import bound as act_bound
from test_fused_act_gpu import oracle_act

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from vllm_quick_amd import _lib
    return _lib


@pytest.fixture(scope="module")
def ops():
    from vllm_quick_amd._C import ops
    return ops


def wide_shapes(L, waves=4):
    """(mt, t, kw) of the private-slab kernel with `waves` wavefronts per workgroup."""
    return [(mt, t, kw) for mt, t, kw in L.built_shapes(2) if kw == waves]


def launch(L, cuda, x, packed, K, N, G, mt, t, kw, splits, ws, bias=None, epilogue=0, out=None, expect=0):
    M = x.shape[0]
    cols = N // 2 if epilogue == 1 else N
    if out is None:
        out = torch.full((M, cols), float("nan"), dtype=torch.float16, device=cuda)
    rc = L.lib.w4a16_gemm_packed_f16_wide_split_cfg(
        x.data_ptr(), packed.packed_w.data_ptr(), packed.packed_sz.data_ptr(), None if bias is None else bias.data_ptr(),
        out.data_ptr(), M, K, N, G, epilogue, mt, t, kw, splits, None if ws is None else ws.data_ptr(),
        0 if ws is None else ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == expect, (rc, L.lib.w4a16_strerror(rc))
    return out


def test_four_wave_shapes_exist(L):
    shapes = wide_shapes(L)
    assert (2, 3, 4) in shapes and (4, 3, 4) in shapes and (2, 2, 4) in shapes


@pytest.mark.parametrize("G", [64, 128])
def test_every_four_wave_shape_with_a_k_split(L, ops, cuda, G):
    # 13 x 32 columns less 16 (ragged against every t and against the 32-column tile), 11 k-blocks (uneven over 2, 3
    # and 4 splits and over the 4 waves of a split), 70 and 133 rows (ragged against 32, 64 and 128-row tiles)
    K, N = 1408, 400
    ws = ops.gemm_workspace(cuda, 8 << 20)
    for M in (70, 133):
        x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=G + M)
        bias = (np.random.default_rng(3).standard_normal(N) * 0.2).astype(np.float16)
        tx, tb = dev_t(x, cuda), dev_t(bias, cuda)
        packed = ops.awq_repack(*[dev_t(a, cuda) for a in (qw, s, qz)])
        want, exact = orc.gemm_faithful(x, qw, s, qz), orc.gemm_exact(x, qw, s, qz)
        for mt, t, kw in wide_shapes(L):
            unsplit = launch(L, cuda, tx, packed, K, N, G, mt, t, kw, 1, None)
            assert_parity(unsplit.cpu().numpy(), want, exact, f"G={G} M={M} ({mt},{t},{kw}) unsplit")
            for splits in (2, 3, 4):
                what = f"G={G} M={M} ({mt},{t},{kw}) splits={splits}"
                need = L.lib.w4a16_wide_workspace_bytes(M, N, 1, 0, mt, t, splits)
                assert 65536 < need <= ws.numel(), what
                y = launch(L, cuda, tx, packed, K, N, G, mt, t, kw, splits, ws)
                assert not torch.isnan(y).any(), what
                assert_parity(y.cpu().numpy(), want, exact, what)
                ulps = 2.0**-9 * np.abs(want.astype(np.float64)) + 1e-6
                assert (np.abs(y.float().cpu().numpy() - unsplit.float().cpu().numpy()) <= ulps).all(), what
                for _ in range(3):  # whoever arrives last, the same bits
                    assert torch.equal(launch(L, cuda, tx, packed, K, N, G, mt, t, kw, splits, ws), y), what
            # bias: an fp16 add on the fp16 result, as `out + bias` (awq.py:163-164)
            yb = launch(L, cuda, tx, packed, K, N, G, mt, t, kw, 2, ws, bias=tb)
            assert torch.equal(yb, launch(L, cuda, tx, packed, K, N, G, mt, t, kw, 2, ws) + tb), (mt, t, kw)
    # every counter went back to zero: nothing but partial sums is left behind the ticket region
    assert int(ws[:65536].view(torch.int32).abs().max().item()) == 0


def test_split_epilogues(L, ops, cuda):
    """silu(gate) * up and residual + linear(x) behind the exchange between workgroups."""
    M, K, N, G = 100, 1536, 448, 128  # 14 + 14 gate / up tiles, 12 k-blocks
    x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=77)
    bias = (np.random.default_rng(5).standard_normal(N) * 0.2).astype(np.float16)
    tx, tb = dev_t(x, cuda), dev_t(bias, cuda)
    packed = ops.awq_repack(*[dev_t(a, cuda) for a in (qw, s, qz)])
    ws = ops.gemm_workspace(cuda, 8 << 20)
    for b_np, b_t in ((None, None), (bias, tb)):
        want, g, u = oracle_act(x, qw, s, qz, b_np)
        tol = act_bound(g, u)
        for mt, t, kw in wide_shapes(L):
            for splits in (1, 2, 3):
                got = launch(L, cuda, tx, packed, K, N, G, mt, t, kw, splits, ws, bias=b_t, epilogue=1)
                got = got.float().cpu().numpy().astype(np.float64)
                assert not np.isnan(got).any(), (mt, t, splits)
                err = np.abs(got - want.astype(np.float64))
                assert (err <= tol).all(), (mt, t, splits, float((err - tol).max()))
    res = (torch.randn(M, N, device=cuda) * 0.5).half()
    for mt, t, kw in wide_shapes(L):
        plain = launch(L, cuda, tx, packed, K, N, G, mt, t, kw, 3, ws, bias=tb)
        out = res.clone()
        launch(L, cuda, tx, packed, K, N, G, mt, t, kw, 3, ws, bias=tb, epilogue=2, out=out)
        assert torch.equal(out, res + plain), (mt, t)  # bit-identical to the two fp16 steps


def test_split_errors(L, ops, cuda):
    M, K, N, G = 64, 1024, 256, 128
    x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=9)
    tx = dev_t(x, cuda)
    packed = ops.awq_repack(*[dev_t(a, cuda) for a in (qw, s, qz)])
    need = L.lib.w4a16_wide_workspace_bytes(M, N, 1, 0, 2, 2, 2)
    assert need == 65536 + 4 * 2 * 4 * 4 * 1024  # tickets, then 4 tiles x 2 splits x 4 owners x 4 KiB
    assert L.lib.w4a16_wide_workspace_bytes(M, N, 1, 0, 2, 2, 1) == 0
    small = torch.zeros(need - 16, dtype=torch.uint8, device=cuda)
    launch(L, cuda, tx, packed, K, N, G, 2, 2, 4, 2, small, expect=-9)  # W4A16_E_WORKSPACE
    launch(L, cuda, tx, packed, K, N, G, 2, 2, 4, 2, None, expect=-4)   # a split needs a workspace
    launch(L, cuda, tx, packed, K, N, G, 2, 2, 8, 2, small, expect=-5)  # 4-wave shapes only
    launch(L, cuda, tx, packed, K, N, G, 2, 2, 4, 9, small, expect=-5)  # at most 8 splits
    launch(L, cuda, tx, packed, K, N, G, 3, 2, 4, 1, None, expect=-5)   # not a built shape
    # splits beyond the k-blocks are trimmed, not an error: 8 blocks over "8 splits" of one block each
    ws = ops.gemm_workspace(cuda, L.lib.w4a16_wide_workspace_bytes(M, N, 1, 0, 2, 2, 8))
    y = launch(L, cuda, tx, packed, K, N, G, 2, 2, 4, 8, ws)
    assert_parity(y.cpu().numpy(), orc.gemm_faithful(x, qw, s, qz), orc.gemm_exact(x, qw, s, qz), "8 splits")


def test_workspace_launch_splits_a_few_column_layer(L, ops, cuda):
    """w4a16_gemm_packed_f16_ws on a tensor-parallel rank's qkv projection (8192 x 1280) at 64 rows: four workgroups
    per tile when it has a workspace, the unsplit launch without; both within tolerance of the oracle."""
    M, K, N, G = 64, 8192, 1280, 128
    arr = (ctypes.c_int * 5)()
    assert L.lib.w4a16_launch_shape_ws(M, K, N, G, 1, 0, arr) == 0
    assert list(arr) == [2, 2, 2, 4, 4], list(arr)  # MFMA32 (mt, t, splits, kw)
    assert L.lib.w4a16_launch_shape(M, K, N, 1, 0, arr) == 0 and arr[0] == 2 and arr[3] == 1
    need = L.lib.w4a16_workspace_bytes(M, K, N, G, 1, 0)
    assert need == L.lib.w4a16_wide_workspace_bytes(M, N, 1, 0, 2, 2, 4) > 0
    x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=12)
    tx = dev_t(x, cuda)
    packed = ops.awq_repack(*[dev_t(a, cuda) for a in (qw, s, qz)])
    ws = ops.gemm_workspace(cuda, need)
    y = ops.awq_gemm_packed(tx, packed, workspace=ws)
    y0 = ops.awq_gemm_packed(tx, packed)
    ref = reference_arithmetic(x, orc.dequantize(qw, s, qz))
    assert_parity_full(y.cpu().numpy(), *ref, what="split")
    assert_parity_full(y0.cpu().numpy(), *ref, what="unsplit")
    assert torch.equal(ops.awq_gemm_packed(tx, packed, workspace=ws), y)
    # a workspace too small for the split: the unsplit launch, not an error
    assert torch.equal(ops.awq_gemm_packed(tx, packed, workspace=ops.gemm_workspace(cuda, 1 << 16)), y0)
    assert int(ws[:65536].view(torch.int32).abs().max().item()) == 0
