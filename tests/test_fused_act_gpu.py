"""GPU parity of the fused gate_up GEMM + silu-and-mul epilogue (W4A16_EPI_SILU_MUL of
w4a16_gemm_packed_f16_ex, called through the C ABI) against the oracle: the reference runs the AWQ GEMM
on the merged [gate | up] weight and then `silu_and_mul` (llama.py:74-80, csrc/activation_kernels.cu:9-34:
silu in fp32 on the fp16 value, rounded to fp16, times `up` in fp16).

Tolerance: the two GEMM results g, u carry the single-GEMM bound (rtol 1e-2 / atol 1e-3 against the
reference-faithful oracle); pushed through out = silu(g) * u (|silu'| <= 1.1) that is
  |d out| <= 1.1 |u| (atol + rtol |g|) + |silu(g)| (atol + rtol |u|)   (+ one fp16 ulp of the result).
Against this library's own unfused route (same arithmetic, possibly another launch shape) the bound is
two fp16 ulps of the operands."""
import numpy as np
import pytest
import torch

from oracle import awq_oracle as orc

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


def oracle_act(x, qw, s, qz, bias=None):
    y = orc.gemm_faithful(x, qw, s, qz)
    if bias is not None:
        y = (torch.from_numpy(y) + torch.from_numpy(bias)).numpy()
    half = y.shape[1] // 2
    g, u = torch.from_numpy(y[:, :half]), torch.from_numpy(y[:, half:])
    gf = g.float()
    sg = (gf / (1.0 + torch.exp(-gf))).half()
    return (sg * u).numpy(), g.numpy().astype(np.float64), u.numpy().astype(np.float64)


def bound(g, u):
    sg = g / (1.0 + np.exp(-g))
    return 1.1 * np.abs(u) * (ATOL + RTOL * np.abs(g)) + np.abs(sg) * (ATOL + RTOL * np.abs(u)) + \
        2.0**-10 * np.abs(sg * u) + 1e-6


def c_abi_act(L, cuda, x, pw, ps, K, N, G, bias=None, cfg=(0, 0, 0, 0), batch=1, xs=0):
    M = x.shape[-2]
    shape = (M, N // 2) if batch == 1 else (batch, M, N // 2)
    out = torch.full(shape, float("nan"), dtype=torch.float16, device=cuda)
    rc = L.lib.w4a16_gemm_packed_f16_ex(x.data_ptr(), xs, pw.data_ptr(), ps.data_ptr(),
                                        None if bias is None else bias.data_ptr(), out.data_ptr(), M * (N // 2), M,
                                        K, N, G, batch, 1, *cfg, torch.cuda.current_stream().cuda_stream)
    L.check(rc, "gemm_ex")
    return out


CASES = [  # M, K, N, G, launch shapes to force (None = the library's choice)
    (1, 512, 256, 128, [None, (1, 2, 16, 0), (1, 2, 8, 0), (1, 4, 8, 0), (1, 2, 4, 0), (1, 4, 4, 0)]),
    (7, 384, 448, 128, [None, (1, 6, 8, 0), (1, 8, 8, 0)]),
    (20, 256, 192, 64, [None, (2, 2, 8, 0), (2, 4, 8, 0), (2, 6, 8, 0), (2, 2, 4, 0)]),
    (70, 640, 416, 32, [None, (4, 3, 2, 4), (4, 3, 4, 2), (2, 3, 2, 4)]),
    (130, 1024, 1376 * 2, 128, [None]),
    # 32x32x16 kernels (a tile = 16 gate + 16 up columns): private slabs (mt, t, 1, -kw), shared slab (mt, t, nw, -1)
    (70, 640, 416, 32, [None, (2, 3, 1, -8), (2, 2, 1, -8), (2, 1, 1, -8), (1, 3, 1, -8), (1, 2, 1, -8), (2, 3, 1, -4),
                        (2, 2, 1, -4), (1, 3, 1, -4), (4, 1, 8, -1), (4, 2, 8, -1), (2, 1, 8, -1),
                        (2, 2, 8, -1), (2, 3, 8, -1)]),
    (40, 256, 192, 64, [None, (2, 3, 1, -8), (1, 2, 1, -8), (2, 2, 1, -4), (2, 2, 8, -1), (4, 1, 8, -1)]),
    (300, 2048, 1376, 128, [None, (2, 3, 1, -8), (4, 2, 8, -1), (4, 1, 8, -1), (2, 3, 8, -1)]),
    (24, 2048, 1376, 128, [None, (1, 2, 1, -8), (1, 3, 1, -4)]),
]


@pytest.mark.parametrize("M,K,N,G,cfgs", CASES)
def test_fused_act_through_c_abi(L, ops, cuda, M, K, N, G, cfgs):
    x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=M + N)
    if K < 2048:  # (longer K: |y| stays O(1), so that the oracle's own fp16 slice rounding stays inside atol)
        s = (s.astype(np.float32) * 2).astype(np.float16)
    bias = (np.random.default_rng(3).standard_normal(N) * 0.2).astype(np.float16)
    tx, tq, ts, tz, tb = [dev_t(a, cuda) for a in (x, qw, s, qz, bias)]
    packed = ops.awq_repack(tq, ts, tz)
    for b_np, b_t in ((None, None), (bias, tb)):
        want, g, u = oracle_act(x, qw, s, qz, b_np)
        tol = bound(g, u)
        y_unfused = ops.awq_gemm_packed(tx, packed, bias=b_t)
        half = N // 2
        unfused = (torch.nn.functional.silu(y_unfused[:, :half].float()).half() * y_unfused[:, half:]).float().cpu().numpy()
        for cfg in cfgs:
            got = c_abi_act(L, cuda, tx, packed.packed_w, packed.packed_sz, K, N, G, b_t, cfg or (0, 0, 0, 0))
            got = got.float().cpu().numpy()
            assert not np.isnan(got).any(), cfg
            err = np.abs(got.astype(np.float64) - want.astype(np.float64))
            assert (err <= tol).all(), (cfg, float((err - tol).max()))
            ulp = 2.0**-9 * (np.abs(u) * 1.1 + np.abs(g / (1 + np.exp(-g))) + np.abs(unfused)) + 1e-6
            assert (np.abs(got - unfused) <= ulp).all(), cfg


def test_fused_act_batched_and_errors(L, ops, cuda):
    E, M, K, N, G = 3, 5, 256, 320, 128
    ws = [orc.make_inputs(M, K, N, G, seed=90 + e) for e in range(E)]
    packed = ops.awq_repack_batch([dev_t(w[1], cuda) for w in ws], [dev_t(w[2], cuda) for w in ws],
                                  [dev_t(w[3], cuda) for w in ws])
    xs = np.stack([w[0] for w in ws])
    y = ops.awq_gemm_packed_silu_and_mul(dev_t(xs, cuda), packed)
    y_shared = ops.awq_gemm_packed_silu_and_mul(dev_t(xs[0], cuda), packed)
    assert y.shape == y_shared.shape == (E, M, N // 2)
    for e, (x, qw, s, qz) in enumerate(ws):
        for got, xin in ((y[e], x), (y_shared[e], xs[0])):
            want, g, u = oracle_act(xin, qw, s, qz)
            err = np.abs(got.float().cpu().numpy().astype(np.float64) - want.astype(np.float64))
            assert (err <= bound(g, u)).all()
    single = ops.awq_gemm_packed_silu_and_mul(dev_t(xs[1], cuda), packed[1])
    assert single.shape == (M, N // 2)
    assert torch.equal(single, y[1]) or (single.float() - y[1].float()).abs().max().item() < 2e-3
    # N must split into two halves of whole tiles; unknown epilogues and unbuilt shapes are refused
    lib = L.lib
    p = packed.packed_w.data_ptr()
    o = y.data_ptr()
    x0 = dev_t(xs[0], cuda)
    assert lib.w4a16_gemm_packed_f16_ex(x0.data_ptr(), 0, p, p, None, o, 0, M, K, 336, G, 1, 1, 0, 0, 0, 0, None) == -5
    assert lib.w4a16_gemm_packed_f16_ex(x0.data_ptr(), 0, p, p, None, o, 0, M, K, N, G, 1, 7, 0, 0, 0, 0, None) == -5
    assert lib.w4a16_gemm_packed_f16_ex(x0.data_ptr(), 0, p, p, None, o, 0, M, K, N, G, 1, 1, 1, 3, 16, 0, None) == -5
    with pytest.raises(ValueError):
        ops.awq_gemm_packed_silu_and_mul(x0, ops.awq_repack(*[dev_t(a, cuda) for a in orc.make_inputs(1, K, 48, G, seed=1)[1:]]))


def test_fused_and_batched_launches_replay_from_a_graph(ops, cuda):
    """The fused-epilogue and batched entry points allocate nothing and synchronise nothing, so they
    capture into a HIP graph (decode batches run as captured graphs in the reference,
    vllm/worker/model_runner.py:658-737); replays must track new inputs."""
    E, M, K, N, G = 2, 4, 256, 192, 64
    ws = [orc.make_inputs(M, K, N, G, seed=30 + e) for e in range(E)]
    batch = ops.awq_repack_batch([dev_t(w[1], cuda) for w in ws], [dev_t(w[2], cuda) for w in ws],
                                 [dev_t(w[3], cuda) for w in ws])
    x = dev_t(ws[0][0], cuda)
    eager = (ops.awq_gemm_packed_silu_and_mul(x, batch[0]), ops.awq_gemm_packed_batched(x, batch),
             ops.awq_gemm_packed_silu_and_mul(x, batch))
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    static_x = x.clone()
    with torch.cuda.stream(stream):
        with torch.cuda.graph(g, stream=stream):
            outs = (ops.awq_gemm_packed_silu_and_mul(static_x, batch[0]), ops.awq_gemm_packed_batched(static_x, batch),
                    ops.awq_gemm_packed_silu_and_mul(static_x, batch))
    g.replay()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(outs, eager))
    x2 = dev_t(ws[1][0], cuda)
    static_x.copy_(x2)
    g.replay()
    torch.cuda.synchronize()
    again = (ops.awq_gemm_packed_silu_and_mul(x2, batch[0]), ops.awq_gemm_packed_batched(x2, batch),
             ops.awq_gemm_packed_silu_and_mul(x2, batch))
    assert all(torch.equal(a, b) for a, b in zip(outs, again))
