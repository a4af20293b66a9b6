"""The hand-counted DMA waits of the two 32x32x16 kernels, checked deterministically.

Those kernels fetch their activation slabs by LDS-DMA and await them with counted `s_waitcnt vmcnt(n)`; a miscount
would show as a rare wrong tile.  libw4a16_check.so is the same library with those two kernels built with
-DW4A16_CHECK_DMA: every piece of LDS is filled with a poison pattern right before the DMA into it is issued, and
every fragment read back is checked for the pattern.  Here every built launch shape of both kernels runs in that
build, must count no fragment that overtook its DMA, and must return the bits the product library returns.  The
negative control: with the build's switch on, the kernels skip the counted waits and read one piece back right
behind the DMA instruction that fills it, which the check must catch (skipping the waits alone need not show: a
slab that arrives early hides a miscount, which is exactly why one would go unnoticed for long).
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import awq_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from vllm_quick_amd import _lib
    return _lib


@pytest.fixture(scope="module")
def chk(L):
    path = os.path.join(os.path.dirname(L.LIB_PATH), "libw4a16_check.so")
    assert os.path.exists(path), f"{path} is missing: build it with __graft_entry__.build()"
    lib = ctypes.CDLL(path)
    for name in ("w4a16_dbg_dma_state_wide", "w4a16_dbg_dma_state_shared"):
        getattr(lib, name).restype = ctypes.c_int
        getattr(lib, name).argtypes = [ctypes.POINTER(ctypes.c_uint), ctypes.c_int]
    lib.w4a16_gemm_packed_f16_wide_cfg.restype = ctypes.c_int
    lib.w4a16_gemm_packed_f16_wide_cfg.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
    return lib


def state(chk, which, set_skip=-1):
    """(fragments that overtook their DMA, skip-waits switch) of one of the two checked kernels."""
    buf = (ctypes.c_uint * 2)()
    fn = chk.w4a16_dbg_dma_state_wide if which == "wide" else chk.w4a16_dbg_dma_state_shared
    torch.cuda.synchronize()
    assert fn(buf, set_skip) == 0
    return int(buf[0]), int(buf[1])


def run(lib, cuda, x, packed, K, N, G, mt, t, kw):
    out = torch.full((x.shape[0], N), float("nan"), dtype=torch.float16, device=cuda)
    rc = lib.w4a16_gemm_packed_f16_wide_cfg(x.data_ptr(), packed.packed_w.data_ptr(), packed.packed_sz.data_ptr(), None,
                                            out.data_ptr(), x.shape[0], K, N, G, mt, t, kw,
                                            torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    return out


def test_no_fragment_overtakes_its_dma_in_any_32x32x16_shape(L, chk, cuda):
    from vllm_quick_amd._C import ops
    K, N, G = 2048, 1184, 128  # 16 k-blocks: every wave of every shape takes several; 37 x 32 columns
    cases = []
    for M in (40, 130, 300):
        x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=M)
        cases.append((torch.from_numpy(x).to(cuda), ops.awq_repack(*[torch.from_numpy(a).to(cuda) for a in (qw, s, qz)])))
    shapes = [("wide", mt, t, kw) for mt, t, kw in L.built_shapes(2)] + \
             [("shared", mt, t, -nw) for mt, t, nw in L.built_shapes(3)]
    assert len(shapes) == 10 + 7
    for which in ("wide", "shared"):
        assert state(chk, which, set_skip=0) == (0, 0)
    for which, mt, t, kw in shapes:
        for tx, packed in cases:
            for rep in range(3):
                got = run(chk, cuda, tx, packed, K, N, G, mt, t, kw)
                want = run(L.lib, cuda, tx, packed, K, N, G, mt, t, kw)
                assert torch.equal(got, want), (which, mt, t, kw, tx.shape[0], rep)
        assert state(chk, which)[0] == 0, f"{which} ({mt},{t},{kw}): a fragment was read before its DMA had landed"
    # negative control: a read right behind its DMA (and no counted waits) is caught, in both kernels
    for which, cfg in (("wide", (2, 3, 4)), ("shared", (4, 2, -8))):
        assert state(chk, which, set_skip=1) == (0, 1)
        run(chk, cuda, cases[1][0], cases[1][1], K, N, G, *cfg)
        caught = state(chk, which)[0]
        assert caught > 0, f"{which}: the check did not notice a read that cannot have seen its DMA's data"
        assert state(chk, which, set_skip=0) == (0, 0)
        got = run(chk, cuda, cases[1][0], cases[1][1], K, N, G, *cfg)
        assert torch.equal(got, run(L.lib, cuda, cases[1][0], cases[1][1], K, N, G, *cfg))
        assert state(chk, which)[0] == 0
