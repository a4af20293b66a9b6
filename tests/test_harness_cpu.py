"""The decode harness's non-GEMM helpers against their textbook forms (the harness itself needs a GPU)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from benchmarks import llama_decode as ld  # noqa: E402


def test_rms_norm_matches_fp32_formula():
    torch.manual_seed(0)
    x, w = torch.randn(5, 256).half(), torch.rand(256).half()
    v = x.float()
    want = (v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + 1e-5)) * w.float()
    got = ld.rms_norm(x, w)
    assert got.dtype == torch.float16
    assert torch.allclose(got.float(), want, rtol=2e-3, atol=2e-3)


def test_rope_neox_matches_two_half_formula():
    torch.manual_seed(1)
    t = torch.randn(3, 6, 1, 32).half()
    ang = torch.rand(16) * 6.28
    cos, sin = ang.cos().half(), ang.sin().half()
    t1, t2 = t[..., :16].float(), t[..., 16:].float()
    want = torch.cat([t1 * cos.float() - t2 * sin.float(), t2 * cos.float() + t1 * sin.float()], -1)
    got = ld.rope_neox(t, *ld.rope_tables(cos, sin))
    assert got.shape == t.shape and got.dtype == torch.float16
    assert torch.allclose(got.float(), want, rtol=2e-3, atol=2e-3)
