"""GPU parity tests of the W4A16 path: the HIP kernels, called through the C ABI of
libw4a16.so, against the CPU oracle and the committed fixtures.

Tolerance (stated by the task for this floating-point path): rtol 1e-2, atol 1e-3 against the
reference-faithful oracle (split-K of 8 with fp16 slices).  Dequantisation is integer/bit work
and must be bit-exact.  At the full 4096 x 11008 size the faithful oracle's own fp16 slice
rounding (up to ~3e-3 absolute, independent of |y|) can by itself exceed atol + rtol|y| on a
handful of near-zero outputs; the full-size tests therefore use the per-element criterion of
tests/_parity.py: such an output must be within the oracle's own rounding bound of the faithful
value and inside rtol/atol of the float64 result, and the kernel must be no further from float64
than the reference arithmetic is.  Everything else is strict.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from _parity import assert_parity as assert_parity_slack
from _parity import reference_arithmetic
from oracle import awq_oracle as orc

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-2, 1e-3
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def L():
    from vllm_quick_amd import _lib
    return _lib


@pytest.fixture(scope="module")
def ops():
    from vllm_quick_amd._C import ops
    return ops


def dev_t(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint16)


def assert_parity(got, faithful, exact, strict=True, what=""):
    assert strict
    got = got.astype(np.float64)
    f = faithful.astype(np.float64)
    e = np.asarray(exact, dtype=np.float64)
    bad = np.abs(got - f) > ATOL + RTOL * np.abs(f)
    assert not bad.any(), f"{what}: {bad.sum()} of {bad.size} outside rtol {RTOL} / atol {ATOL}"
    assert np.abs(got - e).max() <= np.abs(f - e).max() * 1.05 + 2.0**-11, what


class CABI:
    """Thin wrapper that drives libw4a16.so with raw device pointers (what a non-torch caller would do)."""

    def __init__(self, L, cuda):
        self.lib, self.check, self.cuda = L.lib, L.check, cuda

    def stream(self):
        return torch.cuda.current_stream().cuda_stream

    def repack(self, qw, s, qz):
        K, N, G = qw.shape[0], qw.shape[1] * 8, qw.shape[0] // s.shape[0]
        pw = torch.empty(self.lib.w4a16_packed_weight_bytes(K, N), dtype=torch.uint8, device=self.cuda)
        ps = torch.empty(self.lib.w4a16_packed_sz_bytes(K, N, G), dtype=torch.uint8, device=self.cuda)
        self.check(self.lib.w4a16_repack(qw.data_ptr(), s.data_ptr(), qz.data_ptr(), pw.data_ptr(), ps.data_ptr(),
                                         K, N, G, self.stream()), "repack")
        return pw, ps, (K, N, G)

    def gemm_packed(self, x, packed, bias=None, cfg=None):
        pw, ps, (K, N, G) = packed
        M = x.shape[0]
        out = torch.empty((M, N), dtype=torch.float16, device=self.cuda)
        b = None if bias is None else bias.data_ptr()
        if cfg is None:
            rc = self.lib.w4a16_gemm_packed_f16(x.data_ptr(), pw.data_ptr(), ps.data_ptr(), b, out.data_ptr(), M, K, N,
                                                G, self.stream())
        elif cfg[0] == "wide":
            rc = self.lib.w4a16_gemm_packed_f16_wide_cfg(x.data_ptr(), pw.data_ptr(), ps.data_ptr(), b, out.data_ptr(),
                                                         M, K, N, G, cfg[1], cfg[2], cfg[3], self.stream())
        elif len(cfg) == 4:
            rc = self.lib.w4a16_gemm_packed_f16_tiled_cfg(x.data_ptr(), pw.data_ptr(), ps.data_ptr(), b,
                                                          out.data_ptr(), M, K, N, G, cfg[0], cfg[1], cfg[2], cfg[3],
                                                          self.stream())
        else:
            rc = self.lib.w4a16_gemm_packed_f16_cfg(x.data_ptr(), pw.data_ptr(), ps.data_ptr(), b, out.data_ptr(), M,
                                                    K, N, G, cfg[0], cfg[1], cfg[2], self.stream())
        self.check(rc, "gemm_packed")
        return out

    def gemm_stock(self, x, qw, s, qz):
        K, N, G = qw.shape[0], qw.shape[1] * 8, qw.shape[0] // s.shape[0]
        out = torch.empty((x.shape[0], N), dtype=torch.float16, device=self.cuda)
        self.check(self.lib.w4a16_gemm_f16(x.data_ptr(), qw.data_ptr(), s.data_ptr(), qz.data_ptr(), out.data_ptr(),
                                           x.shape[0], K, N, G, 8, self.stream()), "gemm")
        return out

    def dequant(self, qw, s, qz):
        K, N, G = qw.shape[0], qw.shape[1] * 8, qw.shape[0] // s.shape[0]
        out = torch.empty((K, N), dtype=torch.float16, device=self.cuda)
        self.check(self.lib.w4a16_dequantize_f16(qw.data_ptr(), s.data_ptr(), qz.data_ptr(), out.data_ptr(), K, N, G,
                                                 self.stream()), "dequantize")
        return out

    def dequant_packed(self, packed):
        pw, ps, (K, N, G) = packed
        out = torch.empty((K, N), dtype=torch.float16, device=self.cuda)
        self.check(self.lib.w4a16_dequantize_packed_f16(pw.data_ptr(), ps.data_ptr(), out.data_ptr(), K, N, G,
                                                        self.stream()), "dequantize_packed")
        return out


@pytest.fixture(scope="module")
def abi(L, cuda):
    return CABI(L, cuda)


# --------------------------------------------------------------------------
# committed fixtures
# --------------------------------------------------------------------------
def test_golden_cases(abi, cuda):
    g = np.load(os.path.join(GOLDEN, "awq_kernel_cases.npz"))
    names = sorted({k.split("/")[0] for k in g.files if not k.startswith("ka/")})
    assert len(names) == 6
    for name in names:
        x, qw, s, qz = [dev_t(g[f"{name}/{k}"], cuda) for k in ("x", "qweight", "scales", "qzeros")]
        packed = abi.repack(qw, s, qz)
        for label, y in (("packed", abi.gemm_packed(x, packed)), ("stock", abi.gemm_stock(x, qw, s, qz))):
            assert_parity(y.cpu().numpy(), g[f"{name}/faithful"], g[f"{name}/exact"], strict=True,
                          what=f"{name}/{label}")
        if f"{name}/dequant" in g.files:
            assert (bits(abi.dequant(qw, s, qz).cpu().numpy()) == bits(g[f"{name}/dequant"])).all(), name
            assert (bits(abi.dequant_packed(packed).cpu().numpy()) == bits(g[f"{name}/dequant"])).all(), name


def test_known_answer_unit_weights(abi, cuda):
    g = np.load(os.path.join(GOLDEN, "awq_kernel_cases.npz"))
    x, qw, s, qz = [dev_t(g[f"ka/unit_{k}"], cuda) for k in ("x", "qweight", "scales", "qzeros")]
    packed = abi.repack(qw, s, qz)
    assert (abi.gemm_packed(x, packed).cpu().numpy() == g["ka/unit_out"]).all()
    assert (abi.gemm_stock(x, qw, s, qz).cpu().numpy() == g["ka/unit_out"]).all()
    # the single word 0x76543210 with zero 0 / scale 1 dequantises to columns 0,4,1,5,2,6,3,7
    K = 32
    qw1 = dev_t(np.tile(g["ka/word"], (K, 1)), cuda)
    s1 = torch.ones((1, 8), dtype=torch.float16, device=cuda)
    qz1 = torch.zeros((1, 1), dtype=torch.int32, device=cuda)
    w = abi.dequant(qw1, s1, qz1).cpu().numpy()
    assert (w == np.tile(g["ka/word_columns"].astype(np.float16), (K, 1))).all()


# --------------------------------------------------------------------------
# seeded inputs against the oracle: ragged M, every launch shape, odd geometry
# --------------------------------------------------------------------------
@pytest.mark.parametrize("M", [1, 2, 15, 16, 17, 31, 33, 64, 127, 129, 255, 300])
def test_ragged_m(abi, cuda, M):
    K, N, Gs = 384, 1376, 128  # 1376 = a TP-8 shard of 11008; not a multiple of 64
    x, qw, s, qz = orc.make_inputs(M, K, N, Gs, seed=M)
    want, exact = orc.gemm_faithful(x, qw, s, qz), orc.gemm_exact(x, qw, s, qz)
    tx, tqw, ts, tqz = [dev_t(a, cuda) for a in (x, qw, s, qz)]
    y = abi.gemm_packed(tx, abi.repack(tqw, ts, tqz))
    assert_parity(y.cpu().numpy(), want, exact, strict=True, what=f"M={M}")
    # the stock-layout kernel: 16-, 32- and 64-row launch shapes, ragged last row block, ragged last 128 columns
    assert_parity(abi.gemm_stock(tx, tqw, ts, tqz).cpu().numpy(), want, exact, strict=True, what=f"stock M={M}")


def all_built_shapes(L):
    """Every launch shape the library holds, in the form CABI.gemm_packed takes: streaming (mt, t, waves),
    LDS-staged (mt, t, nw, kw), 32x32x16 ("wide", mt, t, kw), shared-slab 32x32x16 ("wide", mt, t, -nw)."""
    cfgs = list(L.built_shapes(0)) + list(L.built_shapes(1))
    cfgs += [("wide", ) + c for c in L.built_shapes(2)]
    cfgs += [("wide", mt, t, -nw) for mt, t, nw in L.built_shapes(3)]
    return cfgs


@pytest.mark.parametrize("G", [32, 64, 128, 256])
def test_every_launch_shape(abi, L, cuda, G):
    CFGS = all_built_shapes(L)
    assert len(CFGS) == 22 + 7 + 10 + 7  # a shape added to the library is a shape tested here
    # 6 row tiles (ragged), 25 column tiles (ragged against every T); 10 k-blocks (uneven over 4 or 8 K-groups),
    # and 16 for the LDS-staged shapes with 8 K-groups, which take whole rounds of 8 blocks only
    M, N = 85, 400
    for K in (1280, 2048):
        x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=G + K)
        want, exact = orc.gemm_faithful(x, qw, s, qz), orc.gemm_exact(x, qw, s, qz)
        tx, tqw, ts, tqz = [dev_t(a, cuda) for a in (x, qw, s, qz)]
        packed = abi.repack(tqw, ts, tqz)
        for cfg in CFGS:
            staged8 = len(cfg) == 4 and cfg[0] != "wide" and cfg[3] == 8
            if K == 1280 and staged8:
                with pytest.raises(ValueError):
                    abi.gemm_packed(tx, packed, cfg=cfg)
                continue
            if K == 2048 and not staged8 and cfg[0] != "wide":
                continue  # covered at 10 blocks
            y = abi.gemm_packed(tx, packed, cfg=cfg)
            assert_parity(y.cpu().numpy(), want, exact, strict=True, what=f"G={G} K={K} cfg={cfg}")
            if len(cfg) == 3 and cfg[0] == 1:
                # the streaming kernel's compact reduction for up to 4 rows (one wave finishes every tile)
                y3 = abi.gemm_packed(tx[:3].contiguous(), packed, cfg=cfg)
                assert_parity(y3.cpu().numpy(), want[:3], exact[:3], strict=True, what=f"G={G} K={K} cfg={cfg} M=3")
        if K == 1280:
            for m in (3, 20, M):  # the three launch shapes of the stock-layout kernel at this group size
                assert_parity(abi.gemm_stock(tx[:m].contiguous(), tqw, ts, tqz).cpu().numpy(), want[:m], exact[:m],
                              strict=True, what=f"G={G} stock M={m}")


@pytest.mark.parametrize("K,N,G", [(96, 40, 32), (160, 8, 32), (32, 16, 32), (224, 72, 32), (4096 + 128, 64, 128)])
def test_odd_geometry(abi, cuda, K, N, G):
    # K not a multiple of 128 (zero-weight padding), N not a multiple of 16 (half-empty last tile)
    for M in (1, 5, 40):
        x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=K + N + M)
        want, exact = orc.gemm_faithful(x, qw, s, qz), orc.gemm_exact(x, qw, s, qz)
        tx, tqw, ts, tqz = [dev_t(a, cuda) for a in (x, qw, s, qz)]
        packed = abi.repack(tqw, ts, tqz)
        assert_parity(abi.gemm_packed(tx, packed).cpu().numpy(), want, exact, strict=True, what=f"{K}x{N} M={M}")
        assert_parity(abi.gemm_stock(tx, tqw, ts, tqz).cpu().numpy(), want, exact, strict=True, what="stock")
        wq = orc.dequantize(qw, s, qz)
        assert (bits(abi.dequant(tqw, ts, tqz).cpu().numpy()) == bits(wq)).all()
        assert (bits(abi.dequant_packed(packed).cpu().numpy()) == bits(wq)).all()


def test_bias_epilogue_is_fp16_add(abi, cuda):
    M, K, N = 9, 256, 264
    x, qw, s, qz = orc.make_inputs(M, K, N, 128, seed=3)
    tx, tqw, ts, tqz = [dev_t(a, cuda) for a in (x, qw, s, qz)]
    packed = abi.repack(tqw, ts, tqz)
    bias = (torch.randn(N, device=cuda) * 0.3).half()
    plain = abi.gemm_packed(tx, packed)
    fused = abi.gemm_packed(tx, packed, bias=bias)
    assert torch.equal(fused, plain + bias)  # same value as the reference's `out + bias` (awq.py:163-164)


# --------------------------------------------------------------------------
# full-size workload (4096 x 11008, group 128): oracle parity and size-independent properties
# --------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full(abi, cuda):
    K, N, G = 4096, 11008, 128
    _, qw, s, qz = orc.make_inputs(1, K, N, G, seed=0)
    tqw, ts, tqz = [dev_t(a, cuda) for a in (qw, s, qz)]
    return dict(K=K, N=N, G=G, qw=qw, s=s, qz=qz, tqw=tqw, ts=ts, tqz=tqz, packed=abi.repack(tqw, ts, tqz),
                w=orc.dequantize(qw, s, qz))


@pytest.mark.parametrize("M", [1, 16, 128])
def test_full_size_parity(abi, cuda, full, M):
    x = orc.make_inputs(M, full["K"], full["N"], full["G"], seed=1000 + M)[0]
    y = abi.gemm_packed(dev_t(x, cuda), full["packed"]).cpu().numpy()
    assert_parity_slack(y, *reference_arithmetic(x, full["w"]), what=f"full M={M}")


@pytest.mark.parametrize("M", [1, 33])
def test_down_proj_shape_parity(abi, cuda, M):
    """11008 -> 4096 (Llama-2-7B down_proj): the long-K layer.  The reference arithmetic's own fp16 slice
    rounding grows with K and |y|; an output that leaves rtol/atol of the faithful oracle must be within
    that rounding's per-element bound of it and inside rtol/atol of the float64 result (tests/_parity.py)."""
    K, N, G = 11008, 4096, 128
    x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=2000 + M)
    packed = abi.repack(dev_t(qw, cuda), dev_t(s, cuda), dev_t(qz, cuda))
    y = abi.gemm_packed(dev_t(x, cuda), packed).cpu().numpy()
    assert (reference_arithmetic(x, orc.dequantize(qw, s, qz))[0] == orc.gemm_faithful(x, qw, s, qz)).all()
    assert_parity_slack(y, *reference_arithmetic(x, orc.dequantize(qw, s, qz)), what=f"down_proj M={M}")


@pytest.mark.parametrize("K,N,M", [
    (1024, 14336, 4), (1024, 14336, 12),    # 769-1535 column tiles: tiles per workgroup by the busiest CU's load
    (1024, 24576, 2), (1024, 24576, 6), (1024, 24576, 12),  # six tiles per CU and more
    (1024, 8192, 48), (1024, 7168, 64),     # 33-64 rows on 448-512 column tiles: streaming shape in one round
    (4096, 4096, 31),                       # LDS-staged two-row-tile shape on a 256-tile layer
    (14336, 256, 2),                        # one or two rows over a very long K
])
def test_default_launch_shape_branches_parity(abi, cuda, K, N, M):
    """The default launch (no shape given) on layers that take the less common branches of the launch-shape
    choice: each must dispatch a built kernel and give the same result as everywhere else."""
    G = 128
    x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=3000 + M + N)
    packed = abi.repack(dev_t(qw, cuda), dev_t(s, cuda), dev_t(qz, cuda))
    y = abi.gemm_packed(dev_t(x, cuda), packed).cpu().numpy()
    assert_parity_slack(y, *reference_arithmetic(x, orc.dequantize(qw, s, qz)), what=f"{K}x{N} M={M}")


@pytest.mark.parametrize("cfg", [("wide", 2, 3, 8), ("wide", 1, 3, 8), ("wide", 2, 2, 4), ("wide", 4, 1, -8), ("wide", 2, 2, -8)])
def test_full_size_wide_kernel(abi, cuda, full, cfg):
    """The 32x32x16 kernel at 4096 x 11008: oracle parity at 100 (ragged) rows, and unit rows x = e_k,
    which must return rows of the dequantised weights bit for bit (the whole k <-> lane <-> word mapping
    of its reading of the packed layout)."""
    x = orc.make_inputs(100, full["K"], full["N"], full["G"], seed=4100)[0]
    y = abi.gemm_packed(dev_t(x, cuda), full["packed"], cfg=cfg).cpu().numpy()
    assert_parity_slack(y, *reference_arithmetic(x, full["w"]), what=f"wide {cfg}")
    ks = [0, 1, 7, 8, 15, 16, 31, 32, 63, 64, 127, 128, 2047, 4095, 1234, 3210, 999, 65, 4000] + list(range(512, 545))
    xe = np.zeros((len(ks), full["K"]), np.float16)
    for i, k in enumerate(ks):
        xe[i, k] = 1.0
    ye = abi.gemm_packed(dev_t(xe, cuda), full["packed"], cfg=cfg).cpu().numpy()
    assert (bits(ye) == bits(full["w"][ks])).all()


def test_full_size_stock_layout_kernel(abi, cuda, full):
    """The stock-layout kernel at 4096 x 11008: oracle parity at 1, 16, 32 and 100 rows, and unit rows, which
    must return rows of the dequantised weights bit for bit (its row -> nibble transposition and its
    column order)."""
    for m in (1, 16, 32, 100):
        x = orc.make_inputs(m, full["K"], full["N"], full["G"], seed=4200 + m)[0]
        y = abi.gemm_stock(dev_t(x, cuda), full["tqw"], full["ts"], full["tqz"]).cpu().numpy()
        assert_parity_slack(y, *reference_arithmetic(x, full["w"]), what=f"stock M={m}")
    ks = [0, 1, 2, 3, 4, 5, 6, 7, 8, 15, 16, 31, 32, 63, 64, 127, 128, 2047, 4095, 1234, 3210, 999, 65, 4000]
    xe = np.zeros((len(ks), full["K"]), np.float16)
    for i, k in enumerate(ks):
        xe[i, k] = 1.0
    for m in (len(ks), 16):
        ye = abi.gemm_stock(dev_t(xe[:m], cuda), full["tqw"], full["ts"], full["tqz"]).cpu().numpy()
        assert (bits(ye) == bits(full["w"][ks[:m]])).all()


def test_full_size_every_32x32x16_shape_repeats_bit_for_bit(abi, L, cuda, full):
    """Many rows of the full-size layer (several workgroups per CU, all of them busy): every launch shape of the two
    32x32x16 kernels must give the same bits on every run (their LDS-DMA is ordered by hand-counted waits: a
    miscount shows up as a sporadic difference, not as a wrong shape) and agree with the default within an fp16 ulp."""
    x = dev_t(orc.make_inputs(1300, full["K"], full["N"], full["G"], seed=4300)[0], cuda)
    ref = abi.gemm_packed(x, full["packed"]).float()
    scale = max(1.0, float(ref.abs().max()))
    shapes = [("wide", ) + c for c in L.built_shapes(2)] + [("wide", mt, t, -nw) for mt, t, nw in L.built_shapes(3)]
    for cfg in shapes:
        first = abi.gemm_packed(x, full["packed"], cfg=cfg)
        assert float((first.float() - ref).abs().max()) <= 2.0**-9 * scale, cfg
        for _ in range(5):
            assert torch.equal(abi.gemm_packed(x, full["packed"], cfg=cfg), first), cfg


def test_full_size_dequantize_bit_exact(abi, full):
    assert (bits(abi.dequant(full["tqw"], full["ts"], full["tqz"]).cpu().numpy()) == bits(full["w"])).all()
    assert (bits(abi.dequant_packed(full["packed"]).cpu().numpy()) == bits(full["w"])).all()


def test_full_size_unit_rows_select_weights(abi, cuda, full):
    # x = e_k: every product but one is zero, so the GEMM must return row k of the dequantised
    # weights bit for bit - checks the whole k <-> lane <-> nibble mapping at full size
    ks = [0, 1, 7, 8, 31, 32, 127, 128, 2047, 4095, 1234, 3210, 999, 64, 65, 4000]
    x = np.zeros((16, full["K"]), np.float16)
    for i, k in enumerate(ks):
        x[i, k] = 1.0
    y = abi.gemm_packed(dev_t(x, cuda), full["packed"]).cpu().numpy()
    assert (bits(y) == bits(full["w"][ks])).all()


def test_full_size_linearity_and_shards(abi, cuda, full):
    K, N, G = full["K"], full["N"], full["G"]
    x1 = dev_t(orc.make_inputs(8, K, N, G, seed=5)[0], cuda)
    x2 = dev_t(orc.make_inputs(8, K, N, G, seed=6)[0], cuda)
    y1, y2 = abi.gemm_packed(x1, full["packed"]).float(), abi.gemm_packed(x2, full["packed"]).float()
    y12 = abi.gemm_packed(x1 + x2, full["packed"]).float()
    assert torch.allclose(y12, y1 + y2, rtol=RTOL, atol=4 * ATOL)
    # column shards (TP column-parallel): each half computed alone equals the matching slice, bit for bit
    cfg = (1, 3, 16)
    whole = abi.gemm_packed(x1, full["packed"], cfg=cfg)
    for r in range(2):
        cs = slice(r * N // 16, (r + 1) * N // 16)
        qw_r = full["tqw"][:, cs].contiguous()
        qz_r = full["tqz"][:, cs].contiguous()
        s_r = full["ts"][:, r * N // 2:(r + 1) * N // 2].contiguous()
        part = abi.gemm_packed(x1, abi.repack(qw_r, s_r, qz_r), cfg=cfg)
        assert torch.equal(part, whole[:, r * N // 2:(r + 1) * N // 2])
    # row shards (TP row-parallel): partial products summed in fp32 match the whole within tolerance
    acc = torch.zeros((8, N), device=cuda)
    for r in range(4):
        ks = slice(r * K // 4, (r + 1) * K // 4)
        gs = slice(r * (K // G) // 4, (r + 1) * (K // G) // 4)
        pk = abi.repack(full["tqw"][ks].contiguous(), full["ts"][gs].contiguous(), full["tqz"][gs].contiguous())
        acc += abi.gemm_packed(x1[:, ks].contiguous(), pk).float()
    assert torch.allclose(acc, whole.float(), rtol=RTOL, atol=4 * ATOL)


# --------------------------------------------------------------------------
# the operator / plugin layer
# --------------------------------------------------------------------------
def test_ops_signatures_and_errors(ops, cuda):
    x, qw, s, qz = [dev_t(a, cuda) for a in orc.make_inputs(4, 256, 128, 128, seed=9)]
    y = ops.awq_gemm(x, qw, s, qz, 8)
    assert y.shape == (4, 128) and y.dtype == torch.float16 and y.device == x.device
    w = ops.awq_dequantize(qw, s, qz, 0, 0, 0)
    assert w.shape == (256, 128) and w.dtype == torch.float16
    assert torch.allclose(y.float(), (x.float() @ w.float()), rtol=RTOL, atol=ATOL)
    assert ops.awq_gemm(x[:0], qw, s, qz, 8).shape == (0, 128)
    with pytest.raises(RuntimeError, match="Half"):
        ops.awq_gemm(x.float(), qw, s, qz, 8)
    with pytest.raises(RuntimeError, match="Int"):
        ops.awq_gemm(x, qw.long(), s, qz, 8)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.awq_gemm(x.cpu(), qw.cpu(), s.cpu(), qz.cpu(), 8)
    with pytest.raises(ValueError):
        ops.awq_gemm(x[:, :128], qw, s, qz, 8)
    with pytest.raises(ValueError):
        ops.awq_gemm(x, qw, s[:, :64], qz, 8)
    # group size 48 is not a multiple of 32 (gemm_kernels.cu:416-417)
    x48, qw48, s48, qz48 = [dev_t(a, cuda) for a in orc.make_inputs(2, 96, 64, 48, seed=1)]
    with pytest.raises(ValueError, match="multiple of 32"):
        ops.awq_gemm(x48, qw48, s48, qz48, 8)
    with pytest.raises(ValueError, match="multiple of 32"):
        ops.awq_dequantize(qw48, s48, qz48, 0, 0, 0)
    # non-contiguous activations are made contiguous, not misread
    xt = x.t().contiguous().t()
    assert not xt.is_contiguous()
    assert torch.equal(ops.awq_gemm(xt, qw, s, qz, 8), y)


def test_ops_repack_cache_tracks_weight_updates(ops, cuda):
    ops.clear_repack_cache()
    x, qw, s, qz = [dev_t(a, cuda) for a in orc.make_inputs(3, 256, 64, 128, seed=2)]
    y0 = ops.awq_gemm(x, qw, s, qz, 8)
    qw2 = dev_t(orc.make_inputs(3, 256, 64, 128, seed=77)[1], cuda)
    qw.copy_(qw2)  # in-place load of new weights into the same storage
    y1 = ops.awq_gemm(x, qw, s, qz, 8)
    want = ops.awq_gemm_stock(x, qw, s, qz)
    assert torch.allclose(y1.float(), want.float(), rtol=RTOL, atol=ATOL)
    assert not torch.equal(y0, y1)


def test_awq_linear_method_matches_reference_fixture(cuda):
    import json
    from vllm_quick_amd.layers.quantization import AWQConfig
    meta = json.load(open(os.path.join(GOLDEN, "awq_reference_python.json")))
    arrays = np.load(os.path.join(GOLDEN, "awq_reference_python.npz"))
    from vllm_quick_amd.layers.quantization.awq import AWQLinearMethod
    cfg = AWQConfig(4, 128, True)
    weights = {k: dev_t(arrays[f"apply/{k}"], cuda) for k in ("qweight", "scales", "qzeros")}
    bias = dev_t(arrays["apply/bias"], cuda)
    # default switch point (fused kernel for every fixture) and the reference's (256 tokens:
    # the m256 fixture then goes through dequantise + matmul)
    for method in (cfg.get_linear_method(), AWQLinearMethod(cfg, fp16_matmul_threshold=256)):
        for label, spec in meta["apply_weights"].items():
            x = dev_t(arrays[f"apply/{label}/x"], cuda)
            y = method.apply_weights(weights, x, bias if spec["bias"] else None)
            assert list(y.shape) == spec["out_shape"] and y.dtype == torch.float16
            want = arrays[f"apply/{label}/out"].astype(np.float32)
            assert np.allclose(y.float().cpu().numpy(), want, rtol=RTOL, atol=2 * ATOL), label
    assert "w4a16_packed" in weights  # the private layout is kept beside the stock tensors


def test_drop_stock_on_gpu(ops, cuda):
    """keep_stock=False: after drop_stock() the stock tensors are gone and the layer answers from the private
    layout alone, bit for bit as before, on the fused route and on the dense large-batch route."""
    from vllm_quick_amd.layers import linear as LL
    from vllm_quick_amd.layers.quantization import AWQConfig
    from vllm_quick_amd.layers.quantization.awq import AWQLinearMethod
    K, N = 512, 384
    _, qw, s, qz = orc.make_inputs(1, K, N, 128, seed=70)
    x = dev_t(orc.make_inputs(300, K, N, 128, seed=71)[0], cuda)
    for threshold in (None, 256):
        method = AWQLinearMethod(AWQConfig(4, 128, True), fp16_matmul_threshold=threshold, keep_stock=False)
        layer = LL.ColumnParallelLinear(K, N, bias=False, params_dtype=torch.float16, linear_method=method).to(cuda)
        for name, arr in (("qweight", qw), ("scales", s), ("qzeros", qz)):
            p = getattr(layer, name)
            p.weight_loader(p, torch.from_numpy(arr).to(cuda))
        before = [layer(x[:m])[0] for m in (5, 300)]
        free0 = torch.cuda.memory_allocated()
        released = method.drop_stock(layer.linear_weights)
        assert released == qw.nbytes + s.nbytes + qz.nbytes and layer.qweight.numel() == 0
        assert torch.cuda.memory_allocated() <= free0 - released + 4096
        after = [layer(x[:m])[0] for m in (5, 300)]
        assert all(torch.equal(a, b) for a, b in zip(before, after))
        fused = method.apply_weights_silu_and_mul(layer.linear_weights, x[:5])
        assert fused.shape == (5, N // 2)


def test_linear_layers_on_gpu(cuda):
    from vllm_quick_amd.layers import linear as LL
    from vllm_quick_amd.layers.quantization import AWQConfig
    method = AWQConfig(4, 128, True).get_linear_method()
    K, Ns = 512, [256, 384]
    ck = [orc.make_inputs(1, K, n, 128, seed=50 + i) for i, n in enumerate(Ns)]
    layer = LL.MergedColumnParallelLinear(K, Ns, bias=False, params_dtype=torch.float16,
                                          linear_method=method).to(cuda)
    for i, (_, qw, s, qz) in enumerate(ck):
        for name, arr in (("qweight", qw), ("scales", s), ("qzeros", qz)):
            p = getattr(layer, name)
            p.weight_loader(p, torch.from_numpy(arr).to(cuda), i)
    x = orc.make_inputs(6, K, Ns[0], 128, seed=60)[0]
    y, ob = layer(dev_t(x, cuda).view(2, 3, K))
    assert ob is None and y.shape == (2, 3, sum(Ns))
    want = np.concatenate([orc.gemm_faithful(x, qw, s, qz) for (_, qw, s, qz) in ck], axis=1)
    assert np.allclose(y.reshape(6, -1).float().cpu().numpy(), want.astype(np.float32), rtol=RTOL, atol=ATOL)


def test_graph_capture_and_replay(ops, cuda):
    x, qw, s, qz = [dev_t(a, cuda) for a in orc.make_inputs(8, 512, 256, 128, seed=4)]
    eager = ops.awq_gemm(x, qw, s, qz, 8)  # warm-up: repacks the weights
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    static_x = x.clone()
    with torch.cuda.stream(stream):
        with torch.cuda.graph(g, stream=stream):
            static_y = ops.awq_gemm(static_x, qw, s, qz, 8)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_y, eager)
    x2 = dev_t(orc.make_inputs(8, 512, 256, 128, seed=5)[0], cuda)
    static_x.copy_(x2)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_y, ops.awq_gemm(x2, qw, s, qz, 8))
