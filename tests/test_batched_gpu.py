"""GPU parity of the batched W4A16 launch (w4a16_gemm_packed_f16_batched, called through the C ABI)
and of the mixture-of-experts block built on it, against the CPU oracle.  Tolerance as for the single
GEMM: rtol 1e-2 / atol 1e-3 against the reference-faithful oracle; the batched launch must also agree
BIT FOR BIT with one single launch per problem (same kernel, same order of additions)."""
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


def make_problems(E, M, K, N, G, seed, shared):
    ws = [orc.make_inputs(M, K, N, G, seed=seed + e) for e in range(E)]
    xs = ws[0][0] if shared else np.stack([w[0] for w in ws])
    return xs, [w[1:] for w in ws]


@pytest.mark.parametrize("M", [1, 7, 16, 33, 70, 300])
@pytest.mark.parametrize("shared", [True, False])
def test_batched_through_c_abi(L, cuda, M, shared):
    E, K, N, G = 3, 384, 208, 128
    xs, ws = make_problems(E, M, K, N, G, 40, shared)
    lib = L.lib
    wb, sb = lib.w4a16_packed_weight_bytes(K, N), lib.w4a16_packed_sz_bytes(K, N, G)
    pw = torch.empty((E, wb), dtype=torch.uint8, device=cuda)
    ps = torch.empty((E, sb), dtype=torch.uint8, device=cuda)
    stream = torch.cuda.current_stream().cuda_stream
    for e, (qw, s, qz) in enumerate(ws):
        tq, ts, tz = dev_t(qw, cuda), dev_t(s, cuda), dev_t(qz, cuda)
        L.check(lib.w4a16_repack(tq.data_ptr(), ts.data_ptr(), tz.data_ptr(), pw[e].data_ptr(), ps[e].data_ptr(),
                                 K, N, G, stream), "repack")
    bias = (np.random.default_rng(1).standard_normal((E, N)) * 0.1).astype(np.float16)
    tb = dev_t(bias, cuda)
    tx = dev_t(xs, cuda)
    out = torch.full((E, M, N), float("nan"), dtype=torch.float16, device=cuda)
    L.check(lib.w4a16_gemm_packed_f16_batched(tx.data_ptr(), 0 if shared else M * K, pw.data_ptr(), ps.data_ptr(),
                                              tb.data_ptr(), out.data_ptr(), M * N, M, K, N, G, E, stream),
            "batched")
    single = torch.empty((E, M, N), dtype=torch.float16, device=cuda)
    for e in range(E):
        xe = tx if shared else tx[e]
        L.check(lib.w4a16_gemm_packed_f16(xe.data_ptr(), pw[e].data_ptr(), ps[e].data_ptr(), tb[e].data_ptr(),
                                          single[e].data_ptr(), M, K, N, G, stream), "single")
    got = out.cpu().numpy()
    for e, (qw, s, qz) in enumerate(ws):
        xe = xs if shared else xs[e]
        want = orc.gemm_faithful(xe, qw, s, qz).astype(np.float32) + bias[e].astype(np.float32)
        assert np.allclose(got[e].astype(np.float32), want, rtol=RTOL, atol=2 * ATOL), (e, M, shared)
    # the default launch shape may differ between batch sizes (more problems, fewer tiles each), and
    # with it the order of the fp32 additions: agreement within one fp16 ulp of the larger result
    a, b = got.astype(np.float32), single.cpu().numpy().astype(np.float32)
    assert np.abs(a - b).max() <= 2.0**-10 * max(1.0, np.abs(b).max())
    if M >= 33:
        # both 32x32x16 kernels as batched launches, whatever the default was: private slabs (mt, t, 1, -kw)
        # and shared slab (mt, t, nw, -1)
        for cfg in ((2, 3, 1, -8), (1, 3, 1, -4), (4, 1, 8, -1), (2, 3, 8, -1), (4, 3, 4, -1)):
            out2 = torch.full((E, M, N), float("nan"), dtype=torch.float16, device=cuda)
            L.check(lib.w4a16_gemm_packed_f16_batched_cfg(tx.data_ptr(), 0 if shared else M * K, pw.data_ptr(),
                                                          ps.data_ptr(), tb.data_ptr(), out2.data_ptr(), M * N, M, K, N,
                                                          G, E, *cfg, stream), "batched_cfg")
            c = out2.cpu().numpy().astype(np.float32)
            assert not np.isnan(c).any(), cfg
            assert np.abs(c - b).max() <= 2.0**-10 * max(1.0, np.abs(b).max()), cfg


def test_batched_out_stride_of_8_byte_alignment(L, cuda):
    """Problems whose outputs start 8 bytes (not 16) apart from a 16-byte grid: the out stride only has to be a multiple
    of 4 elements.  The shared-slab kernel stores 16 bytes per lane, the others 8: both must land where a per-problem
    launch on the same addresses puts them."""
    E, M, K, N, G = 3, 70, 384, 208, 128
    xs, ws = make_problems(E, M, K, N, G, 77, True)
    lib = L.lib
    wb, sb = lib.w4a16_packed_weight_bytes(K, N), lib.w4a16_packed_sz_bytes(K, N, G)
    pw = torch.empty((E, wb), dtype=torch.uint8, device=cuda)
    ps = torch.empty((E, sb), dtype=torch.uint8, device=cuda)
    stream = torch.cuda.current_stream().cuda_stream
    for e, (qw, s, qz) in enumerate(ws):
        tq, ts, tz = dev_t(qw, cuda), dev_t(s, cuda), dev_t(qz, cuda)
        L.check(lib.w4a16_repack(tq.data_ptr(), ts.data_ptr(), tz.data_ptr(), pw[e].data_ptr(), ps[e].data_ptr(),
                                 K, N, G, stream), "repack")
    tx = dev_t(xs, cuda)
    stride = M * N + 4
    for cfg in ((4, 1, 8, -1), (2, 3, 8, -1), (4, 3, 4, -1), (2, 3, 1, -8), None):
        out = torch.full((E * stride,), float("nan"), dtype=torch.float16, device=cuda)
        if cfg is None:  # the library's launch shape
            L.check(lib.w4a16_gemm_packed_f16_batched(tx.data_ptr(), 0, pw.data_ptr(), ps.data_ptr(), None, out.data_ptr(),
                                                      stride, M, K, N, G, E, stream), "batched")
        else:
            L.check(lib.w4a16_gemm_packed_f16_batched_cfg(tx.data_ptr(), 0, pw.data_ptr(), ps.data_ptr(), None,
                                                          out.data_ptr(), stride, M, K, N, G, E, *cfg, stream), "batched_cfg")
        got = out.cpu().numpy()
        for e, (qw, s, qz) in enumerate(ws):
            want = orc.gemm_faithful(xs, qw, s, qz).astype(np.float32)
            block = got[e * stride:e * stride + M * N].reshape(M, N).astype(np.float32)
            assert np.allclose(block, want, rtol=RTOL, atol=2 * ATOL), (cfg, e)
            assert np.isnan(got[e * stride + M * N:(e + 1) * stride]).all(), (cfg, e, "padding between problems written")


def test_batched_argument_errors(L, cuda):
    lib = L.lib
    t = torch.zeros(4096, dtype=torch.uint8, device=cuda)
    p = t.data_ptr()
    assert lib.w4a16_gemm_packed_f16_batched(p, 0, p, p, None, p, 16, 1, 128, 16, 128, 0, None) == -5  # batch 0
    assert lib.w4a16_gemm_packed_f16_batched(p, 4, p, p, None, p, 16, 1, 128, 16, 128, 2, None) == -5  # x stride % 8
    assert lib.w4a16_gemm_packed_f16_batched(p, 0, p, p, None, p, 8, 1, 128, 16, 128, 2, None) == -5   # out stride < M*N
    assert lib.w4a16_gemm_packed_f16_batched(p, 0, p, p, None, p, 16, 1, 128, 12, 128, 2, None) == -1  # N % 8
    assert lib.w4a16_gemm_packed_f16_batched(None, 0, p, p, None, p, 16, 1, 128, 16, 128, 2, None) == -4


def test_ops_batched_wrapper(ops, cuda):
    E, M, K, N, G = 4, 5, 256, 96, 64
    xs, ws = make_problems(E, M, K, N, G, 70, shared=False)
    packed = ops.awq_repack_batch([dev_t(w[0], cuda) for w in ws], [dev_t(w[1], cuda) for w in ws],
                                  [dev_t(w[2], cuda) for w in ws])
    assert packed.batch == E and (packed.K, packed.N, packed.G) == (K, N, G)
    y = ops.awq_gemm_packed_batched(dev_t(xs, cuda), packed)
    assert y.shape == (E, M, N)
    for e, (qw, s, qz) in enumerate(ws):
        want = orc.gemm_faithful(xs[e], qw, s, qz).astype(np.float32)
        assert np.allclose(y[e].float().cpu().numpy(), want, rtol=RTOL, atol=ATOL)
        # a problem of the batch is an ordinary packed weight
        y1 = ops.awq_gemm_packed(dev_t(xs[e], cuda), packed[e])
        assert np.allclose(y1.float().cpu().numpy(), want, rtol=RTOL, atol=ATOL)
    with pytest.raises(ValueError):
        ops.awq_gemm_packed_batched(dev_t(xs[:2], cuda), packed)
    with pytest.raises(RuntimeError):
        ops.awq_gemm_packed_batched(dev_t(xs, cuda).float(), packed)


def test_mixtral_block_on_gpu(cuda):
    """Three batched launches against the reference's per-expert loop (mixtral_quant.py:133-160),
    restated on the oracle."""
    from test_experts_cpu import fill, reference_block
    from vllm_quick_amd.layers import experts as X
    from vllm_quick_amd.layers.quantization import AWQConfig
    E, top_k, H, I = 4, 2, 128, 224
    method = AWQConfig(4, 32, True).get_linear_method()
    moe = X.MixtralMoE(E, top_k, H, I, linear_method=method, params_dtype=torch.float16)
    ck, gate = fill(moe, seed=9)
    moe = moe.to(cuda)
    x = (np.random.default_rng(2).standard_normal((2, 4, H)) * 0.5).astype(np.float16)
    with torch.inference_mode():
        y = moe(dev_t(x, cuda))
        moe.batched = False
        y_loop = moe(dev_t(x, cuda))
    want = reference_block(x.reshape(8, H), gate, ck, moe.expert_indicies, top_k).astype(np.float32)
    # Two chained GEMMs with outputs of magnitude ~10: the fp16 roundings of the intermediate and of
    # the reference's own slices are absolute errors proportional to that magnitude, so the absolute
    # part of the tolerance is scaled by it (the 1e-3 of the single GEMM assumes |y| = O(1)).
    scale = max(1.0, float(np.abs(want).max()))
    for got in (y, y_loop):
        assert got.shape == (2, 4, H)
        g = got.reshape(8, H).float().cpu().numpy()
        assert (np.abs(g - want) <= RTOL * np.abs(want) + ATOL * scale).all()
    # the two routes run the same kernels on the same numbers up to the final fp32-vs-fp16 expert sum
    assert (y.float() - y_loop.float()).abs().max().item() <= 2.0**-9 * scale


@pytest.mark.parametrize("epilogue", [0, 1])
@pytest.mark.parametrize("M", [1, 4, 40])
def test_masked_batch_through_c_abi(L, cuda, M, epilogue):
    """w4a16_gemm_packed_f16_masked: switched-on problems bit for bit what the unmasked launch writes,
    switched-off problems' output untouched; the mask is device data."""
    E, K, N, G = 5, 256, 224, 128
    xs, ws = make_problems(E, M, K, N, G, 70, True)
    lib = L.lib
    wb, sb = lib.w4a16_packed_weight_bytes(K, N), lib.w4a16_packed_sz_bytes(K, N, G)
    pw = torch.empty((E, wb), dtype=torch.uint8, device=cuda)
    ps = torch.empty((E, sb), dtype=torch.uint8, device=cuda)
    stream = torch.cuda.current_stream().cuda_stream
    for e, (qw, s, qz) in enumerate(ws):
        tq, ts, tz = dev_t(qw, cuda), dev_t(s, cuda), dev_t(qz, cuda)
        L.check(lib.w4a16_repack(tq.data_ptr(), ts.data_ptr(), tz.data_ptr(), pw[e].data_ptr(), ps[e].data_ptr(),
                                 K, N, G, stream), "repack")
    tx = dev_t(xs, cuda)
    cols = N // 2 if epilogue else N
    full = torch.empty((E, M, cols), dtype=torch.float16, device=cuda)
    L.check(lib.w4a16_gemm_packed_f16_ex(tx.data_ptr(), 0, pw.data_ptr(), ps.data_ptr(), None, full.data_ptr(),
                                         M * cols, M, K, N, G, E, epilogue, 0, 0, 0, 0, stream), "ex")
    for on in ([1, 0, 0, 1, 0], [0, 0, 0, 0, 0], [1, 1, 1, 1, 1], [0, 7, 0, 0, -1]):
        mask = torch.tensor(on, dtype=torch.int32, device=cuda)
        hint = 0  # the hint may change the launch shape, hence the order of additions: compared without it
        out = torch.full((E, M, cols), 123.0, dtype=torch.float16, device=cuda)
        L.check(lib.w4a16_gemm_packed_f16_masked(tx.data_ptr(), 0, pw.data_ptr(), ps.data_ptr(), None, out.data_ptr(),
                                                 M * cols, M, K, N, G, E, epilogue, mask.data_ptr(), hint, stream), "masked")
        torch.cuda.synchronize()
        for e in range(E):
            if on[e]:
                assert torch.equal(out[e], full[e]), (on, e)
            else:
                assert bool((out[e] == 123.0).all()), (on, e)
    # no mask = every problem; a misaligned mask pointer is refused
    out = torch.empty((E, M, cols), dtype=torch.float16, device=cuda)
    L.check(lib.w4a16_gemm_packed_f16_masked(tx.data_ptr(), 0, pw.data_ptr(), ps.data_ptr(), None, out.data_ptr(),
                                             M * cols, M, K, N, G, E, epilogue, None, 0, stream), "masked, no mask")
    torch.cuda.synchronize()
    assert torch.equal(out, full)
    assert lib.w4a16_gemm_packed_f16_masked(tx.data_ptr(), 0, pw.data_ptr(), ps.data_ptr(), None, out.data_ptr(),
                                            M * cols, M, K, N, G, E, epilogue, mask.data_ptr() + 2, 0, stream) == -6


def test_mixtral_block_skips_unselected_experts_on_gpu(cuda):
    """One to three tokens: the block with the per-expert switch (computed on the device) returns the very
    bits of the block that runs every expert; also from a captured graph with fresh tokens per replay."""
    from test_experts_cpu import fill
    from vllm_quick_amd.layers import experts as X
    from vllm_quick_amd.layers.quantization import AWQConfig
    E, top_k, H, I = 8, 2, 128, 224
    moe = X.MixtralMoE(E, top_k, H, I, linear_method=AWQConfig(4, 32, True).get_linear_method(),
                       params_dtype=torch.float16)
    fill(moe, seed=19)
    moe = moe.to(cuda)
    rng = np.random.default_rng(3)
    with torch.inference_mode():
        for tokens in (1, 2, 3):
            x = dev_t((rng.standard_normal((1, tokens, H)) * 0.5).astype(np.float16), cuda)
            moe.sparse_max_tokens = 4
            sparse = moe(x)
            w = moe.routing(x.view(-1, H))
            assert int((w != 0).any(0).sum()) <= top_k * tokens < E
            moe.sparse_max_tokens = 0
            dense = moe(x)
            assert torch.equal(sparse, dense)
        moe.sparse_max_tokens = 4
        xg = torch.zeros((1, 1, H), dtype=torch.float16, device=cuda)
        stream = torch.cuda.Stream(device=cuda)
        with torch.cuda.stream(stream):
            moe(xg)
            stream.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                yg = moe(xg)
            for _ in range(3):
                fresh = dev_t((rng.standard_normal((1, 1, H)) * 0.5).astype(np.float16), cuda)
                xg.copy_(fresh)
                g.replay()
                stream.synchronize()
                moe.sparse_max_tokens = 0
                assert torch.equal(yg, moe(fresh))
                moe.sparse_max_tokens = 4


def framework_route(logits, top_k, first, n):
    """mixtral_quant.py:137-150 with framework operators (what `MixtralMoE.routing` runs)."""
    p = torch.nn.functional.softmax(logits, dim=1, dtype=torch.float)
    p, sel = torch.topk(p, top_k, dim=-1)
    p /= p.sum(dim=-1, keepdim=True)
    dense = torch.zeros(logits.shape, dtype=torch.float, device=logits.device).scatter_(1, sel, p)
    return dense[:, first:first + n]


@pytest.mark.parametrize("tokens", [1, 3, 16, 300, 1000])
def test_moe_route_against_framework_operators(ops, cuda, tokens):
    """w4a16_moe_route_f16: the routing weights of the reference's block in one launch.  Same selection as
    softmax / topk / renormalise / scatter, weights to the last bits of fp32 (the kernel's exp and summation
    order are its own), the problem mask = 'some token selected this expert'."""
    gen = torch.Generator(device=cuda).manual_seed(tokens)
    for E, top_k, first, n in ((8, 2, 0, 8), (8, 2, 4, 4), (8, 2, 6, 2), (16, 4, 0, 16), (64, 2, 32, 8), (3, 3, 0, 3)):
        logits = (torch.randn((tokens, E), generator=gen, device=cuda) * 2).half()
        want = framework_route(logits, top_k, first, n)
        w, mask = ops.moe_route(logits, top_k, first, n, want_mask=True)
        assert w.shape == want.shape and w.dtype == torch.float32 and mask.dtype == torch.int32
        assert torch.equal(w != 0, want != 0), (E, top_k, first, n)   # (random fp16 logits: ties are rare; see below)
        assert torch.allclose(w, want, rtol=2e-6, atol=1e-9)
        assert torch.equal(mask.bool(), (want != 0).any(0))
        w2, none = ops.moe_route(logits, top_k, first, n)
        assert none is None and torch.equal(w2, w)
    # ties go to the lower index, and a row's weights sum to one
    flat = torch.zeros((2, 8), dtype=torch.float16, device=cuda)
    w, _ = ops.moe_route(flat, 2)
    assert torch.equal(w, torch.tensor([[0.5, 0.5, 0, 0, 0, 0, 0, 0]] * 2, device=cuda))
    with pytest.raises(ValueError):
        ops.moe_route(flat, 9)             # more selections than experts
    with pytest.raises(ValueError):
        ops.moe_route(flat, 2, 4, 8)       # local range past the last expert
    with pytest.raises(RuntimeError):
        ops.moe_route(flat.float(), 2)     # fp16 logits only (the gate is an fp16 linear layer)


@pytest.mark.parametrize("tokens", [1, 4, 33])
def test_moe_combine_against_framework_operators(ops, cuda, tokens):
    """w4a16_moe_combine_f16: bit for bit the framework sequence (scale in fp32, round to fp16, sum over the
    experts in fp32, round once) for top-2 routing, where the sum has two terms; rows of experts no token
    selected are never read (they hold NaN here)."""
    gen = torch.Generator(device=cuda).manual_seed(7 + tokens)
    E, H = 8, 4096
    logits = (torch.randn((tokens, E), generator=gen, device=cuda) * 2).half()
    w, mask = ops.moe_route(logits, 2, want_mask=True)
    y = (torch.randn((E, tokens, H), generator=gen, device=cuda) * 3).half()
    want = torch.mul(y, w.t().unsqueeze(-1)).to(torch.float16).sum(0, dtype=torch.float32).to(torch.float16)
    assert torch.equal(ops.moe_combine(y, w), want)
    poisoned = y.clone()
    poisoned[w.t() == 0] = float("nan")   # every (expert, token) row the token did not select
    assert torch.equal(ops.moe_combine(poisoned, w), want)
    # four selected experts: the fp32 sum of four terms in expert order, within an fp16 ulp of any other order
    w4, _ = ops.moe_route(logits, 4)
    got = ops.moe_combine(y, w4).float()
    want4 = torch.mul(y, w4.t().unsqueeze(-1)).to(torch.float16).sum(0, dtype=torch.float32)
    assert ((got - want4).abs() <= 2.0**-10 * want4.abs() + 1e-7).all()
    with pytest.raises(ValueError):
        ops.moe_combine(y, w[:, :4])
    with pytest.raises(RuntimeError):
        ops.moe_combine(y.float(), w)


def test_mixtral_block_fused_routing_matches_operator_route(cuda):
    """The block with routing and the weighted sum as one launch each against the same block on framework
    operators: same GEMM launches, routing weights equal to a few fp32 ulps, hence outputs within an fp16 ulp."""
    from test_experts_cpu import fill
    from vllm_quick_amd.layers import experts as X
    from vllm_quick_amd.layers.quantization import AWQConfig
    E, top_k, H, I = 8, 2, 128, 224
    moe = X.MixtralMoE(E, top_k, H, I, linear_method=AWQConfig(4, 32, True).get_linear_method(),
                       params_dtype=torch.float16)
    fill(moe, seed=23)
    moe = moe.to(cuda)
    rng = np.random.default_rng(5)
    with torch.inference_mode():
        for tokens in (1, 3, 16, 70):
            x = dev_t((rng.standard_normal((1, tokens, H)) * 0.5).astype(np.float16), cuda)
            assert moe.fused_routing
            fused = moe(x)
            moe.fused_routing = False
            plain = moe(x)
            moe.fused_routing = True
            assert not torch.isnan(fused).any()
            tol = 2.0**-9 * plain.float().abs() + 2.0**-9 * plain.float().abs().max() * 1e-2 + 1e-6
            assert ((fused.float() - plain.float()).abs() <= tol).all(), tokens
