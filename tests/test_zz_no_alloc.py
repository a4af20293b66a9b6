"""The entry point that allocates nothing: w4a16_gemm_packed_f16_decode_static runs the K-split decode launch on
the library's own file-scope device array instead of a caller's workspace.  Called through the C ABI and compared,
bit for bit (same kernel, same split, same summation order), with w4a16_gemm_packed_f16_decode_cfg on a workspace
the test allocated: two calls in a row, the second on a smaller input, so that whatever the first call left in the
array is what the second one finds."""
import ctypes

import numpy as np
import pytest
import torch

from _parity import assert_parity, reference_arithmetic
from oracle import awq_oracle as orc

pytestmark = pytest.mark.gpu


def dev_t(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


@pytest.fixture(scope="module")
def L():
    from vllm_quick_amd import _lib
    return _lib


def call(L, cuda, static, x, pw, ps, K, N, G, t, waves, splits, ws=None, epilogue=0, expect=0):
    M = x.shape[0]
    cols = N // 2 if epilogue == 1 else N
    out = torch.full((M, cols), float("nan"), dtype=torch.float16, device=cuda)
    common = (x.data_ptr(), 0, pw.data_ptr(), ps.data_ptr(), None, out.data_ptr(), 0, M, K, N, G, 1, epilogue, t, waves,
              0, splits)
    s = torch.cuda.current_stream().cuda_stream
    if static:
        rc = L.lib.w4a16_gemm_packed_f16_decode_static(*common, s)
    else:
        rc = L.lib.w4a16_gemm_packed_f16_decode_cfg(*common, ws.data_ptr(), ws.numel(), s)
    assert rc == expect, (rc, L.lib.w4a16_strerror(rc))
    return out


def test_static_scratch_entry_point_matches_the_workspace_one(L, cuda):
    assert L.lib.w4a16_static_scratch_bytes() == 1 << 20
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device=cuda)
    # first a larger input, then a smaller one: fewer workgroup columns, fewer splits, fewer rows
    cases = [(4, 14336, 512, 1, 16, 3, 0), (1, 2048, 512, 2, 8, 2, 0), (3, 4096, 640, 2, 16, 2, 1)]
    for rep in range(2):
        for M, K, N, t, waves, splits, epilogue in cases:
            x, qw, s, qz = orc.make_inputs(M, K, N, 128, seed=M + K + N + rep)
            tx, tqw, ts, tqz = [dev_t(a, cuda) for a in (x, qw, s, qz)]
            pw = torch.empty(L.lib.w4a16_packed_weight_bytes(K, N), dtype=torch.uint8, device=cuda)
            ps = torch.empty(L.lib.w4a16_packed_sz_bytes(K, N, 128), dtype=torch.uint8, device=cuda)
            assert L.lib.w4a16_repack(tqw.data_ptr(), ts.data_ptr(), tqz.data_ptr(), pw.data_ptr(), ps.data_ptr(), K, N,
                                      128, torch.cuda.current_stream().cuda_stream) == 0
            got = call(L, cuda, True, tx, pw, ps, K, N, 128, t, waves, splits, epilogue=epilogue)
            want = call(L, cuda, False, tx, pw, ps, K, N, 128, t, waves, splits, ws, epilogue=epilogue)
            torch.cuda.synchronize()
            assert not torch.isnan(got).any()
            assert torch.equal(got, want), (rep, M, K, N)
            if epilogue == 0:
                # rtol 1e-2 / atol 1e-3 against the reference-faithful arithmetic, its own fp16 slice rounding
                # (which grows with K) bounded per element
                assert_parity(got.cpu().numpy(), *reference_arithmetic(x, orc.dequantize(qw, s, qz)), what=f"{K}x{N}")
            # t = 0: the launch shape the library picks when it has a workspace
            auto = call(L, cuda, True, tx, pw, ps, K, N, 128, 0, 0, 0, epilogue=epilogue)
            assert (auto.float() - want.float()).abs().max().item() <= 2e-3 * max(1.0, want.float().abs().max().item())


def test_static_scratch_entry_point_rejects_what_does_not_fit(L, cuda):
    M, K, N = 1, 1024, 8192  # 512 workgroup columns x 8 splits x 1 KiB of partial sums: 4 MiB
    assert L.lib.w4a16_decode_workspace_bytes(K, N, 1, 0, 1, 8, 8) > L.lib.w4a16_static_scratch_bytes()
    x, qw, s, qz = orc.make_inputs(M, K, N, 128, seed=1)
    tx, tqw, ts, tqz = [dev_t(a, cuda) for a in (x, qw, s, qz)]
    pw = torch.empty(L.lib.w4a16_packed_weight_bytes(K, N), dtype=torch.uint8, device=cuda)
    ps = torch.empty(L.lib.w4a16_packed_sz_bytes(K, N, 128), dtype=torch.uint8, device=cuda)
    assert L.lib.w4a16_repack(tqw.data_ptr(), ts.data_ptr(), tqz.data_ptr(), pw.data_ptr(), ps.data_ptr(), K, N, 128,
                              torch.cuda.current_stream().cuda_stream) == 0
    out = call(L, cuda, True, tx, pw, ps, K, N, 128, 1, 8, 8, expect=-9)
    torch.cuda.synchronize()
    assert torch.isnan(out).all()  # nothing was launched
    assert b"scratch" in L.lib.w4a16_strerror(-9)
    # 65 rows are not a decode-kernel problem: no launch shape of this entry point serves them
    x65 = dev_t(orc.make_inputs(65, K, N, 128, seed=2)[0], cuda)
    call(L, cuda, True, x65, pw, ps, K, N, 128, 0, 0, 0, expect=-5)
    # the fused activation takes tile PAIRS: one tile (and any odd count) is no launch shape, with a K split or
    # without, and is refused before anything is computed from "tiles per workgroup"
    K2, N2 = 1024, 512
    x1, qw2, s2, qz2 = orc.make_inputs(1, K2, N2, 128, seed=3)
    tx1, tqw2, ts2, tqz2 = [dev_t(a, cuda) for a in (x1, qw2, s2, qz2)]
    pw2 = torch.empty(L.lib.w4a16_packed_weight_bytes(K2, N2), dtype=torch.uint8, device=cuda)
    ps2 = torch.empty(L.lib.w4a16_packed_sz_bytes(K2, N2, 128), dtype=torch.uint8, device=cuda)
    assert L.lib.w4a16_repack(tqw2.data_ptr(), ts2.data_ptr(), tqz2.data_ptr(), pw2.data_ptr(), ps2.data_ptr(), K2, N2,
                              128, torch.cuda.current_stream().cuda_stream) == 0
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device=cuda)
    for t in (1, 3):
        for splits in (1, 2):
            assert L.lib.w4a16_decode_workspace_bytes(K2, N2, 1, 1, t, 8, splits) == 0
            call(L, cuda, True, tx1, pw2, ps2, K2, N2, 128, t, 8, splits, epilogue=1, expect=-5)
            call(L, cuda, False, tx1, pw2, ps2, K2, N2, 128, t, 8, splits, ws, epilogue=1, expect=-5)
    # and a split that fits runs
    got = call(L, cuda, True, tx, pw, ps, K, N, 128, 2, 8, 2)
    f = orc.gemm_faithful(x, qw, s, qz).astype(np.float64)
    assert (np.abs(got.float().cpu().numpy() - f) <= 1e-3 + 1e-2 * np.abs(f)).all()
