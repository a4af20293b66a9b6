"""GPU parity of the decode kernel (up to 16 rows; w4a16_gemm_packed_f16_decode_cfg / _ws of the C ABI) against the
oracle: every built (tiles, waves) shape at both prefetch depths, with and without a split of K over workgroups,
every group size the kernel admits, bias / fused-activation / add-to-out epilogues, batched and masked launches.

Tolerance as for every GEMM here: rtol 1e-2 / atol 1e-3 against the reference-faithful oracle.  A launch with a K
split sums the same fp32 products in another order than the unsplit one; the two must agree to two fp16 ulps, and a
split launch must return the same bits every time (the partial sums are added in split order, not in arrival order).
"""
import numpy as np
import pytest
import torch

from oracle import awq_oracle as orc
from test_fused_act_gpu import bound as act_bound
from test_fused_act_gpu import oracle_act

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-2, 1e-3


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


def decode_shapes(L, act=False):
    """(tiles, waves) of the decode kernel; the fused activation takes the even tile counts (gate / up pairs)."""
    return [(t, w) for t, w, _ in L.built_shapes(4) if (t % 2 == 0 or not act)]


def splittable(t, act=False):
    """The K split is built for workgroups of up to 4 tiles (or gate / up tile pairs)."""
    return (t // 2 if act else t) <= 4


def assert_parity(got, want, exact, what):
    got, f = got.astype(np.float64), want.astype(np.float64)
    bad = np.abs(got - f) > ATOL + RTOL * np.abs(f)
    assert not bad.any(), f"{what}: {bad.sum()} of {bad.size} outside rtol {RTOL} / atol {ATOL}"
    assert np.abs(got - exact).max() <= np.abs(f - exact).max() * 1.05 + 2.0**-11, what


def launch(L, cuda, x, packed, K, N, G, t, waves, depth=0, splits=1, ws=None, bias=None, epilogue=0, batch=1,
           x_stride=0, out=None, expect=0):
    """w4a16_gemm_packed_f16_decode_cfg with raw pointers; returns the output tensor."""
    M = x.shape[-2]
    cols = N // 2 if epilogue == 1 else N
    if out is None:
        shape = (M, cols) if batch == 1 else (batch, M, cols)
        out = torch.full(shape, float("nan"), dtype=torch.float16, device=cuda)
    rc = L.lib.w4a16_gemm_packed_f16_decode_cfg(
        x.data_ptr(), x_stride, packed.packed_w.data_ptr(), packed.packed_sz.data_ptr(),
        None if bias is None else bias.data_ptr(), out.data_ptr(), M * cols, M, K, N, G, batch, epilogue, t, waves,
        depth, splits, None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(),
        torch.cuda.current_stream().cuda_stream)
    assert rc == expect, (rc, L.lib.w4a16_strerror(rc))
    return out


def test_decode_shapes_are_enumerated(L):
    shapes = decode_shapes(L)
    assert len(shapes) == 11 == len(L.built_shapes(4))  # a shape added to the library is a shape tested here
    assert (3, 16) in shapes and (4, 8) in shapes and (8, 8) in shapes and (2, 4) in shapes
    assert set(decode_shapes(L, act=True)) < set(shapes)


@pytest.mark.parametrize("G", [32, 64, 128, 256])
def test_every_decode_shape(L, ops, cuda, G):
    # 25 column tiles (ragged against every t); 10 and 22 k-blocks (uneven over 8 or 16 waves and over 2 or 3 splits)
    N = 400
    ws = ops.gemm_workspace(cuda, 1 << 20)
    for K in (1280, 2816):
        for M in (1, 3, 4, 5, 9, 16):
            x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=G + K + M)
            want, exact = orc.gemm_faithful(x, qw, s, qz), orc.gemm_exact(x, qw, s, qz)
            bias = (np.random.default_rng(M).standard_normal(N) * 0.2).astype(np.float16)
            tx, tb = dev_t(x, cuda), dev_t(bias, cuda)
            packed = ops.awq_repack(*[dev_t(a, cuda) for a in (qw, s, qz)])
            for t, waves in decode_shapes(L):
                unsplit = None
                for depth in (1, 2):
                    for splits in ((1, 2, 3) if M <= 4 and splittable(t) else (1, )):  # (up to 4 rows and 4 tiles)
                        what = f"G={G} K={K} M={M} t={t} waves={waves} depth={depth} splits={splits}"
                        y = launch(L, cuda, tx, packed, K, N, G, t, waves, depth, splits, ws)
                        assert_parity(y.cpu().numpy(), want, exact, what)
                        if splits == 1 and depth == 1:
                            unsplit = y
                        elif splits == 1:
                            assert torch.equal(y, unsplit), what  # the prefetch depth does not touch the arithmetic
                        else:
                            ulps = 2.0**-9 * np.abs(want.astype(np.float64)) + 1e-6
                            assert (np.abs(y.float().cpu().numpy() - unsplit.float().cpu().numpy()) <= ulps).all(), what
                # bias: an fp16 add on the fp16 result, as `out + bias` (awq.py:163-164)
                sp = 2 if M <= 4 and splittable(t) else 1
                yb = launch(L, cuda, tx, packed, K, N, G, t, waves, 0, sp, ws, bias=tb)
                y2 = launch(L, cuda, tx, packed, K, N, G, t, waves, 0, sp, ws)
                assert torch.equal(yb, y2 + tb), (G, K, M, t, waves)
    # every counter went back to zero: nothing but partial sums is left behind the ticket region
    assert int(ws[:65536].view(torch.int32).abs().max().item()) == 0


@pytest.mark.parametrize("G", [32, 64, 128])
def test_two_row_tiles_17_to_32_rows(L, ops, cuda, G):
    """17-32 rows: two 16-row tiles through the LDS ring, one column tile x 16 waves; ragged columns, k-blocks
    uneven over the waves (and enough of them for the oldest waves to take a block from the youngest), bias and
    add-to-out, whatever depth is asked for."""
    N = 400
    for K in (1280, 2816, 4736):
        for M in (17, 23, 32):
            x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=G + K + M)
            want, exact = orc.gemm_faithful(x, qw, s, qz), orc.gemm_exact(x, qw, s, qz)
            bias = (np.random.default_rng(M).standard_normal(N) * 0.2).astype(np.float16)
            tx, tb = dev_t(x, cuda), dev_t(bias, cuda)
            packed = ops.awq_repack(*[dev_t(a, cuda) for a in (qw, s, qz)])
            for t, waves in ((1, 16), ):
                what = f"G={G} K={K} M={M} t={t}"
                y = launch(L, cuda, tx, packed, K, N, G, t, waves, 0, 1, None)
                assert_parity(y.cpu().numpy(), want, exact, what)
                assert torch.equal(launch(L, cuda, tx, packed, K, N, G, t, waves, 2, 1, None), y), what
                yb = launch(L, cuda, tx, packed, K, N, G, t, waves, 0, 1, None, bias=tb)
                assert torch.equal(yb, y + tb), what
                res = dev_t((np.random.default_rng(K).standard_normal((M, N))).astype(np.float16), cuda)
                ya = launch(L, cuda, tx, packed, K, N, G, t, waves, 0, 1, None, epilogue=2, out=res.clone())
                assert torch.equal(ya, res + y), what
    # the library takes this path on a narrow layer
    assert "gemm_decode_kernel<1,16>" in L.launch_shape(24, 4096, 4096)


def test_split_launches_repeat_bit_for_bit_and_share_a_workspace(L, ops, cuda):
    """One workspace, never cleared again, serves launches of different shapes back to back, and a split launch
    returns the same bits however its workgroups are scheduled."""
    ws = ops.gemm_workspace(cuda, 2 << 20)
    cases = []
    for K, N, M in ((8192, 1280, 1), (4096, 1376, 4), (2048, 512, 2)):
        x, qw, s, qz = orc.make_inputs(M, K, N, 128, seed=K + N)
        cases.append((dev_t(x, cuda), ops.awq_repack(*[dev_t(a, cuda) for a in (qw, s, qz)]), K, N,
                      orc.gemm_faithful(x, qw, s, qz), orc.gemm_exact(x, qw, s, qz)))
    first = {}
    for rep in range(6):
        for i, (tx, packed, K, N, want, exact) in enumerate(cases):
            for splits in (2, 3, 4):
                y = launch(L, cuda, tx, packed, K, N, 128, 1, 16, 0, splits, ws)
                key = (i, splits)
                if key not in first:
                    first[key] = y
                    assert_parity(y.cpu().numpy(), want, exact, f"{K}x{N} splits={splits}")
                else:
                    assert torch.equal(y, first[key]), (key, rep)


def test_workspace_errors_and_sizes(L, ops, cuda):
    K, N, G, M = 1024, 256, 128, 2
    x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=5)
    tx = dev_t(x, cuda)
    packed = ops.awq_repack(*[dev_t(a, cuda) for a in (qw, s, qz)])
    need = L.lib.w4a16_decode_workspace_bytes(K, N, 1, 0, 1, 8, 2)
    assert need == 65536 + 16 * 2 * 1024  # ticket region, then 16 workgroup columns x 2 splits x 1 KiB
    assert L.lib.w4a16_decode_workspace_bytes(K, N, 1, 0, 1, 8, 1) == 0
    small = torch.zeros(need - 16, dtype=torch.uint8, device=cuda)
    launch(L, cuda, tx, packed, K, N, G, 1, 8, 0, 2, small, expect=-9)   # W4A16_E_WORKSPACE
    launch(L, cuda, tx, packed, K, N, G, 1, 8, 0, 2, None, expect=-4)    # a split needs a workspace
    launch(L, cuda, tx, packed, K, N, G, 1, 8, 0, 9, small, expect=-5)   # at most 8 splits
    launch(L, cuda, tx, packed, K, N, G, 5, 8, 0, 1, None, expect=-5)    # not a built shape
    launch(L, cuda, tx, packed, K, N, G, 6, 8, 0, 2, small, expect=-5)   # the K split takes up to 4 tiles a workgroup
    launch(L, cuda, tx, packed, K, N, G, 1, 8, 3, 1, None, expect=-5)    # depth is 0, 1 or 2
    x5 = dev_t(orc.make_inputs(5, K, N, G, seed=6)[0], cuda)
    launch(L, cuda, x5, packed, K, N, G, 1, 8, 0, 2, ws_ok := torch.zeros(need, dtype=torch.uint8, device=cuda),
           expect=-5)                                                      # the K split takes up to 4 rows
    x17 = dev_t(orc.make_inputs(17, K, N, G, seed=6)[0], cuda)
    launch(L, cuda, x17, packed, K, N, G, 1, 8, 0, 1, None, expect=-5)   # more than 16 rows: 1 tile x 16 waves only
    launch(L, cuda, x17, packed, K, N, G, 3, 16, 0, 1, None, expect=-5)
    launch(L, cuda, x17, packed, K, N, G, 2, 16, 0, 1, None, expect=-5)
    launch(L, cuda, x17, packed, K, N, G, 1, 16, 0, 1, None, epilogue=1, expect=-5)  # ... and the plain GEMM only
    x33 = dev_t(orc.make_inputs(33, K, N, G, seed=6)[0], cuda)
    launch(L, cuda, x33, packed, K, N, G, 1, 16, 0, 1, None, expect=-5)  # more than 32 rows
    assert L.lib.w4a16_strerror(-9).startswith(b"workspace too small")
    # splits beyond the k-blocks are trimmed, not an error: 8 blocks over "8 splits" of one block each
    ws = ops.gemm_workspace(cuda, L.lib.w4a16_decode_workspace_bytes(K, N, 1, 0, 1, 8, 8))
    y = launch(L, cuda, tx, packed, K, N, G, 1, 8, 0, 8, ws)
    assert_parity(y.cpu().numpy(), orc.gemm_faithful(x, qw, s, qz), orc.gemm_exact(x, qw, s, qz), "8 splits")


@pytest.mark.parametrize("M", [1, 4, 7, 16])
def test_decode_fused_activation_and_add(L, ops, cuda, M):
    K, N, G = 1536, 448, 128  # 14 + 14 gate / up tiles, 12 k-blocks
    x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=40 + M)
    bias = (np.random.default_rng(4).standard_normal(N) * 0.2).astype(np.float16)
    tx, tb = dev_t(x, cuda), dev_t(bias, cuda)
    packed = ops.awq_repack(*[dev_t(a, cuda) for a in (qw, s, qz)])
    ws = ops.gemm_workspace(cuda, 1 << 20)
    for b_np, b_t in ((None, None), (bias, tb)):
        want, g, u = oracle_act(x, qw, s, qz, b_np)
        tol = act_bound(g, u)
        # this library's own unfused route: the same arithmetic up to the launch shape's summation order
        y = ops.awq_gemm_packed(tx, packed, bias=b_t, cfg=(1, 2, 8))
        unfused = (torch.nn.functional.silu(y[:, :N // 2].float()).half() * y[:, N // 2:]).float().cpu().numpy()
        ulp = 2.0**-9 * (np.abs(u) * 1.1 + np.abs(g / (1 + np.exp(-g))) + np.abs(unfused)) + 1e-6
        for t, waves in decode_shapes(L):
            if t % 2:
                launch(L, cuda, tx, packed, K, N, G, t, waves, 0, 1, None, bias=b_t, epilogue=1, expect=-5)
        for t, waves in decode_shapes(L, act=True):
            for splits in ((1, 2, 3) if M <= 4 and splittable(t, act=True) else (1, )):
                got = launch(L, cuda, tx, packed, K, N, G, t, waves, 0, splits, ws, bias=b_t, epilogue=1)
                got = got.float().cpu().numpy().astype(np.float64)
                assert not np.isnan(got).any()
                err = np.abs(got - want.astype(np.float64))
                assert (err <= tol).all(), (t, waves, splits, float((err - tol).max()))
                assert (np.abs(got - unfused) <= ulp).all(), (t, waves, splits)
    # residual + linear(x) in one launch: bit-identical to the two fp16 steps
    res = (torch.randn(M, N, device=cuda) * 0.5).half()
    for t, waves in decode_shapes(L):
        for splits in ((1, 2) if M <= 4 and splittable(t) else (1, )):
            plain = launch(L, cuda, tx, packed, K, N, G, t, waves, 0, splits, ws, bias=tb)
            out = res.clone()
            launch(L, cuda, tx, packed, K, N, G, t, waves, 0, splits, ws, bias=tb, epilogue=2, out=out)
            assert torch.equal(out, res + plain), (t, waves, splits)


def test_decode_batched_and_masked(L, ops, cuda):
    E, M, K, N, G = 5, 3, 1024, 272, 128
    probs = [orc.make_inputs(M, K, N, G, seed=70 + e) for e in range(E)]
    packed = ops.awq_repack_batch([dev_t(p[1], cuda) for p in probs], [dev_t(p[2], cuda) for p in probs],
                                  [dev_t(p[3], cuda) for p in probs])
    xs = dev_t(np.stack([p[0] for p in probs]), cuda)
    ws = ops.gemm_workspace(cuda, 1 << 20)
    for t, waves in ((1, 16), (2, 8), (3, 8)):
        for splits in (1, 2):
            for shared in (False, True):
                y = launch(L, cuda, xs[0] if shared else xs, packed, K, N, G, t, waves, 0, splits, ws, batch=E,
                           x_stride=0 if shared else M * K)
                for e, (x, qw, s, qz) in enumerate(probs):
                    xin = probs[0][0] if shared else x
                    assert_parity(y[e].cpu().numpy(), orc.gemm_faithful(xin, qw, s, qz), orc.gemm_exact(xin, qw, s, qz),
                                  f"problem {e} t={t} waves={waves} splits={splits} shared={shared}")
    # the library's own choice with a mask and a workspace: switched-off problems keep what `out` held
    mask = torch.tensor([1, 0, 1, 0, 1], dtype=torch.int32, device=cuda)
    out = torch.full((E, M, N), 7.0, dtype=torch.float16, device=cuda)
    full = launch(L, cuda, xs, packed, K, N, G, 1, 16, 0, 1, None, batch=E, x_stride=M * K)
    rc = L.lib.w4a16_gemm_packed_f16_masked(xs.data_ptr(), M * K, packed.packed_w.data_ptr(),
                                            packed.packed_sz.data_ptr(), None, out.data_ptr(), M * N, M, K, N, G, E, 0,
                                            mask.data_ptr(), 3, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    for e in range(E):
        if e % 2 == 0:
            assert (out[e].float() - full[e].float()).abs().max().item() <= 2e-3
        else:
            assert (out[e] == 7.0).all()


def test_decode_batched_nine_rows(L, ops, cuda):
    E, M, K, N, G = 3, 9, 1024, 272, 128
    probs = [orc.make_inputs(M, K, N, G, seed=80 + e) for e in range(E)]
    packed = ops.awq_repack_batch([dev_t(p[1], cuda) for p in probs], [dev_t(p[2], cuda) for p in probs],
                                  [dev_t(p[3], cuda) for p in probs])
    xs = dev_t(np.stack([p[0] for p in probs]), cuda)
    for t, waves in ((1, 16), (2, 8)):
        y = launch(L, cuda, xs, packed, K, N, G, t, waves, 0, 1, None, batch=E, x_stride=M * K)
        for e, (x, qw, s, qz) in enumerate(probs):
            assert_parity(y[e].cpu().numpy(), orc.gemm_faithful(x, qw, s, qz), orc.gemm_exact(x, qw, s, qz), f"problem {e}")


def test_library_choice_with_a_workspace(L, ops, cuda):
    """A narrow layer with a long K at a few rows is split over K once a workspace is there, served by the decode
    kernel without one; both within tolerance, the default entry points unchanged."""
    import ctypes
    K, N, G = 14336, 512, 128
    buf = (ctypes.c_int * 5)()
    assert L.lib.w4a16_launch_shape(1, K, N, 1, 0, buf) == 0 and list(buf) == [4, 1, 1, 16, 1]
    assert L.lib.w4a16_launch_shape_ws(1, K, N, G, 1, 0, buf) == 0 and buf[0] == 4 and buf[4] == 3
    assert L.lib.w4a16_launch_shape_ws(64, K, N, G, 1, 0, buf) == 0 and buf[0] != 4
    assert L.lib.w4a16_workspace_bytes(1, K, N, G, 1, 0) > 65536
    assert L.lib.w4a16_workspace_bytes(1, 4096, 11008, G, 1, 0) == 0   # enough column tiles: no split
    assert L.lib.w4a16_workspace_bytes(1, 8192, 1280, G, 1, 0) == 0    # K too short for a split to pay
    # 64 rows of this layer: the private-slab 32x32x16 kernel splits K over workgroups too (tickets + partial sums)
    assert L.lib.w4a16_workspace_bytes(64, K, N, G, 1, 0) > 65536
    assert L.lib.w4a16_workspace_bytes(512, K, N, G, 1, 0) == 0  # many rows: nothing is split
    ws = ops.gemm_workspace(cuda)
    for M in (1, 4):
        x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=11 + M)
        tx = dev_t(x, cuda)
        packed = ops.awq_repack(*[dev_t(a, cuda) for a in (qw, s, qz)])
        want, exact = orc.gemm_faithful(x, qw, s, qz), orc.gemm_exact(x, qw, s, qz)
        y_ws = ops.awq_gemm_packed(tx, packed, workspace=ws)
        y = ops.awq_gemm_packed(tx, packed)
        for got, what in ((y_ws, "with workspace"), (y, "default")):
            assert_parity(got.cpu().numpy(), want, exact, f"{what} M={M}")
        # a workspace that is too small is ignored, not an error
        tiny = torch.zeros(64, dtype=torch.uint8, device=cuda)
        assert torch.equal(ops.awq_gemm_packed(tx, packed, workspace=tiny), y)
    with pytest.raises(RuntimeError):
        ops.awq_gemm_packed(tx, packed, workspace=torch.zeros(1 << 20, dtype=torch.uint8))  # a host tensor


def test_split_launch_replays_from_a_graph(L, ops, cuda):
    K, N, G, M = 4096, 640, 128, 2
    x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=21)
    packed = ops.awq_repack(*[dev_t(a, cuda) for a in (qw, s, qz)])
    ws = ops.gemm_workspace(cuda, 1 << 20)
    tx = dev_t(x, cuda)
    out = torch.zeros((M, N), dtype=torch.float16, device=cuda)
    eager = launch(L, cuda, tx, packed, K, N, G, 1, 16, 0, 3, ws)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        launch(L, cuda, tx, packed, K, N, G, 1, 16, 0, 3, ws, out=out)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            launch(L, cuda, tx, packed, K, N, G, 1, 16, 0, 3, ws, out=out)
        for _ in range(3):
            out.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, eager)
        x2 = orc.make_inputs(M, K, N, G, seed=22)[0]
        tx.copy_(dev_t(x2, cuda))
        g.replay()
        torch.cuda.synchronize()
    assert_parity(out.cpu().numpy(), orc.gemm_faithful(x2, qw, s, qz), orc.gemm_exact(x2, qw, s, qz), "replay")


def test_linear_method_split_k_option(ops, cuda):
    """AWQLinearMethod(split_k=True) routes up to 4 tokens through the stream's workspace; same results within the
    tolerance, more rows and the default method untouched."""
    from vllm_quick_amd.layers.quantization import AWQConfig
    from vllm_quick_amd.layers.quantization.awq import AWQLinearMethod
    K, N, G = 14336, 512, 128
    cfg = AWQConfig(4, G, True)
    plain, split = AWQLinearMethod(cfg), AWQLinearMethod(cfg, split_k=True)
    for M in (1, 4, 9):
        x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=31 + M)
        weights = {"qweight": dev_t(qw, cuda), "scales": dev_t(s, cuda), "qzeros": dev_t(qz, cuda)}
        tx = dev_t(x, cuda)
        y0, y1 = plain.apply_weights(weights, tx), split.apply_weights(dict(weights), tx)
        if M > 4:
            assert torch.equal(y0, y1)
        else:
            ulps = 2.0**-9 * y0.float().abs() + 1e-6
            assert ((y0.float() - y1.float()).abs() <= ulps).all()
    ws = ops.stream_workspace(cuda)
    assert ws is not None and ws.numel() == 1 << 20 and ops.stream_workspace(cuda) is ws
    assert int(ws[:65536].view(torch.int32).abs().max().item()) == 0


@pytest.mark.parametrize("N,shape_1,shape_16", [(13824, "<4,8>", "<4,8>"), (22016, "<6,8>", "<6,8>"), (28672, "<4,4>", "<4,4>")])
def test_wide_layers_take_their_decode_shapes(L, ops, cuda, N, shape_1, shape_16):
    """Three to six column tiles per CU run one workgroup per CU (4 or 6 tiles x 8 waves), more than six run
    4-wave workgroups of 4 tiles: the library's own choice on layers that wide (K kept short for the oracle),
    plain and with the fused activation, at 1, 4, 9 and 16 rows."""
    K, G = 1024, 128
    assert L.launch_shape(1, K, N) == "gemm_decode_kernel" + shape_1
    assert L.launch_shape(16, K, N) == "gemm_decode_kernel" + shape_16
    x16, qw, s, qz = orc.make_inputs(16, K, N, G, seed=N)
    packed = ops.awq_repack(*[dev_t(a, cuda) for a in (qw, s, qz)])
    for M in (1, 4, 9, 16):
        x = x16[:M]
        want, exact = orc.gemm_faithful(x, qw, s, qz), orc.gemm_exact(x, qw, s, qz)
        y = ops.awq_gemm_packed(dev_t(x, cuda), packed)
        assert_parity(y.cpu().numpy(), want, exact, f"N={N} M={M}")
        assert L.launch_shape(M, K, N, 1, 1).startswith("gemm_decode_kernel")
        got = ops.awq_gemm_packed_silu_and_mul(dev_t(x, cuda), packed).float().cpu().numpy().astype(np.float64)
        ref, g, u = oracle_act(x, qw, s, qz, None)
        err = np.abs(got - ref.astype(np.float64))
        assert (err <= act_bound(g, u)).all(), (N, M, float((err - act_bound(g, u)).max()))
