"""The compiled PyTorch binding `vllm_quick_amd._C_native` (pybind11, csrc/torch_binding.cpp): the module a
vLLM build would import in place of `vllm._C` for the two AWQ operators (csrc/pybind.cpp:52-55 of the
reference).  It must run the matrix-instruction kernels (bit-identical to the packed-layout entry point),
keep the reference's signatures and exception types, and be capturable in a HIP graph after a warm-up."""
import numpy as np
import pytest
import torch

from _parity import assert_parity, reference_arithmetic
from oracle import awq_oracle as orc

pytestmark = pytest.mark.gpu


def dev_t(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


@pytest.fixture(scope="module")
def native():
    from vllm_quick_amd import _C_native
    return _C_native


@pytest.fixture(scope="module")
def ops():
    from vllm_quick_amd._C import ops
    return ops


@pytest.mark.parametrize("M,K,N,G", [(1, 4096, 4096, 128), (16, 1024, 1376, 128), (128, 4096, 11008, 128), (300, 512, 264, 64)])
def test_awq_gemm_runs_the_packed_kernels(native, ops, cuda, M, K, N, G):
    x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=M + N)
    tx, tq, ts, tz = [dev_t(a, cuda) for a in (x, qw, s, qz)]
    before = native.stock_kernel_calls()
    y = native.ops.awq_gemm(tx, tq, ts, tz, 8)
    assert y.shape == (M, N) and y.dtype == torch.float16 and y.device == tx.device
    assert native.stock_kernel_calls() == before  # served by the private-layout kernels
    assert torch.equal(y, ops.awq_gemm_packed(tx, ops.awq_repack(tq, ts, tz)))
    assert_parity(y.cpu().numpy(), *reference_arithmetic(x, orc.dequantize(qw, s, qz)), what=f"native {M}x{K}x{N}")
    w = native.ops.awq_dequantize(tq, ts, tz, 0, 0, 0)
    assert (w.cpu().numpy().view(np.uint16) == orc.dequantize(qw, s, qz).view(np.uint16)).all()


def test_signatures_errors_and_cache(native, cuda):
    x, qw, s, qz = [dev_t(a, cuda) for a in orc.make_inputs(4, 256, 128, 128, seed=9)]
    y = native.ops.awq_gemm(x, qw, s, qz, 8)
    assert native.ops.awq_gemm(x[:0], qw, s, qz, 8).shape == (0, 128)
    with pytest.raises(RuntimeError, match="Half"):
        native.ops.awq_gemm(x.float(), qw, s, qz, 8)
    with pytest.raises(RuntimeError, match="Int"):
        native.ops.awq_gemm(x, qw.long(), s, qz, 8)
    with pytest.raises(RuntimeError, match="GPU"):
        native.ops.awq_gemm(x.cpu(), qw.cpu(), s.cpu(), qz.cpu(), 8)
    with pytest.raises(ValueError):
        native.ops.awq_gemm(x[:, :128], qw, s, qz, 8)
    with pytest.raises(ValueError):
        native.ops.awq_gemm(x, qw, s[:, :64], qz, 8)
    x48, qw48, s48, qz48 = [dev_t(a, cuda) for a in orc.make_inputs(2, 96, 64, 48, seed=1)]
    with pytest.raises(ValueError, match="multiple of 32"):  # gemm_kernels.cu:416-417
        native.ops.awq_gemm(x48, qw48, s48, qz48, 8)
    with pytest.raises(ValueError, match="multiple of 32"):
        native.ops.awq_dequantize(qw48, s48, qz48, 0, 0, 0)
    # non-contiguous activations are made contiguous, not misread
    xt = x.t().contiguous().t()
    assert not xt.is_contiguous() and torch.equal(native.ops.awq_gemm(xt, qw, s, qz, 8), y)
    # the table follows in-place weight loading: same storage, new contents -> re-packed
    n0 = native.repack_count()
    assert torch.equal(native.ops.awq_gemm(x, qw, s, qz, 8), y) and native.repack_count() == n0
    qw.copy_(dev_t(orc.make_inputs(4, 256, 128, 128, seed=77)[1], cuda))
    y2 = native.ops.awq_gemm(x, qw, s, qz, 8)
    assert native.repack_count() == n0 + 1 and not torch.equal(y2, y)
    want = orc.gemm_faithful(x.cpu().numpy(), qw.cpu().numpy(), s.cpu().numpy(), qz.cpu().numpy())
    assert np.allclose(y2.float().cpu().numpy(), want.astype(np.float32), rtol=1e-2, atol=1e-3)


def test_graph_capture_and_replay(native, cuda):
    x, qw, s, qz = [dev_t(a, cuda) for a in orc.make_inputs(8, 512, 256, 128, seed=4)]
    eager = native.ops.awq_gemm(x, qw, s, qz, 8)  # warm-up: re-orders the weights
    torch.cuda.synchronize()
    before = native.stock_kernel_calls()
    stream = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    static_x = x.clone()
    with torch.cuda.stream(stream):
        with torch.cuda.graph(g, stream=stream):
            static_y = native.ops.awq_gemm(static_x, qw, s, qz, 8)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_y, eager) and native.stock_kernel_calls() == before
    x2 = dev_t(orc.make_inputs(8, 512, 256, 128, seed=5)[0], cuda)
    static_x.copy_(x2)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_y, native.ops.awq_gemm(x2, qw, s, qz, 8))
    # a first call under capture cannot re-order the weights: it is served from the stock layout and counted
    _, qw3, s3, qz3 = [dev_t(a, cuda) for a in orc.make_inputs(8, 512, 256, 128, seed=6)]
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        with torch.cuda.graph(g2, stream=stream):
            y3 = native.ops.awq_gemm(static_x, qw3, s3, qz3, 8)
    g2.replay()
    torch.cuda.synchronize()
    assert native.stock_kernel_calls() == before + 1
    assert torch.allclose(y3.float(), native.ops.awq_gemm(static_x, qw3, s3, qz3, 8).float(), rtol=1e-2, atol=1e-3)


def test_python_binding_counts_and_warns_when_the_stock_kernel_serves(ops, cuda):
    """The ctypes mirror of the two operators behaves like the compiled one: warmed-up layers replay from a
    graph on the packed kernels; a layer first seen under capture is served by the stock-layout kernel,
    which is counted and warned about (once per process)."""
    import warnings
    x, qw, s, qz = [dev_t(a, cuda) for a in orc.make_inputs(8, 512, 256, 128, seed=14)]
    eager = ops.awq_gemm(x, qw, s, qz, 8)  # warm-up
    torch.cuda.synchronize()
    before = ops.stock_kernel_calls()
    stream = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream), torch.cuda.graph(g, stream=stream):
        y = ops.awq_gemm(x, qw, s, qz, 8)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, eager) and ops.stock_kernel_calls() == before
    x2, qw2, s2, qz2 = [dev_t(a, cuda) for a in orc.make_inputs(8, 512, 256, 128, seed=15)]
    g2 = torch.cuda.CUDAGraph()
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        with torch.cuda.stream(stream), torch.cuda.graph(g2, stream=stream):
            y2 = ops.awq_gemm(x2, qw2, s2, qz2, 8)
    g2.replay()
    torch.cuda.synchronize()
    assert ops.stock_kernel_calls() == before + 1
    if before == 0:
        assert any("stock-layout kernel" in str(w.message) for w in seen)
    want = orc.gemm_faithful(*[t.cpu().numpy() for t in (x2, qw2, s2, qz2)])
    assert np.allclose(y2.float().cpu().numpy(), want.astype(np.float32), rtol=1e-2, atol=1e-3)


@pytest.mark.parametrize("binding", ["native", "ctypes"])
def test_weights_without_version_counters_are_never_served_stale(native, ops, cuda, binding):
    """Tensors created under torch.inference_mode() keep no version counter, so an in-place reload of them cannot be
    told from no change: neither binding keeps a packed copy for them (every call re-packs, said once).  Writes
    through `param.data` (the reference's loaders) do not touch the counter of a normal tensor either: there the
    caller clears the cache after the load, as the integration notes require."""
    import warnings
    gemm = native.ops.awq_gemm if binding == "native" else ops.awq_gemm
    clear = native.clear_repack_cache if binding == "native" else ops.clear_repack_cache
    M, K, N, G = 4, 256, 128, 128
    x_np, qw_np, s_np, qz_np = orc.make_inputs(M, K, N, G, seed=31)
    _, qw2_np, s2_np, qz2_np = orc.make_inputs(M, K, N, G, seed=32)

    def want(qw_, s_, qz_):
        return orc.gemm_faithful(x_np, qw_, s_, qz_).astype(np.float32)

    with torch.inference_mode():
        x, qw, s, qz = [dev_t(a, cuda) for a in (x_np, qw_np, s_np, qz_np)]
        assert qw.is_inference() and s.is_inference()
        with warnings.catch_warnings(record=True) as seen:
            warnings.simplefilter("always")
            y1 = gemm(x, qw, s, qz, 8)
            n0 = native.repack_count()
            y1b = gemm(x, qw, s, qz, 8)
        assert torch.equal(y1, y1b)
        if binding == "native":
            assert native.repack_count() == n0 + 1  # re-packed again: nothing was kept
        else:
            assert any("inference-mode" in str(w.message) for w in seen) or ops._warned_untracked
        assert np.allclose(y1.float().cpu().numpy(), want(qw_np, s_np, qz_np), rtol=1e-2, atol=1e-3)
        # reload in place, same storage, under inference mode: the next call must see the new weights
        qw.copy_(dev_t(qw2_np, cuda)), s.copy_(dev_t(s2_np, cuda)), qz.copy_(dev_t(qz2_np, cuda))
        y2 = gemm(x, qw, s, qz, 8)
        assert np.allclose(y2.float().cpu().numpy(), want(qw2_np, s2_np, qz2_np), rtol=1e-2, atol=1e-3)
        assert not torch.equal(y1, y2)
    # normal tensors written through .data (counter untouched): right again once the cache was cleared
    x, qw, s, qz = [dev_t(a, cuda) for a in (x_np, qw_np, s_np, qz_np)]
    y1 = gemm(x, qw, s, qz, 8)
    v = qw._version
    qw.data.copy_(dev_t(qw2_np, cuda)), s.data.copy_(dev_t(s2_np, cuda)), qz.data.copy_(dev_t(qz2_np, cuda))
    assert qw._version == v  # which is why the cache cannot see it
    clear()
    y2 = gemm(x, qw, s, qz, 8)
    assert np.allclose(y2.float().cpu().numpy(), want(qw2_np, s2_np, qz2_np), rtol=1e-2, atol=1e-3)
    assert not torch.equal(y1, y2)
