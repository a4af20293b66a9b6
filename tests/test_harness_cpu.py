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


def test_one_rank_rehearsal_shards_layers_and_copies_instead_of_reducing():
    """parallel_state.rehearse_one_rank(N): layers built inside shard as one rank of N would (linear.py:191-201,
    516-526 of the reference: narrow on the output / input dimension), the all-reduce is a copy, and nothing of it
    outlives the block."""
    import pytest
    from vllm_quick_amd.layers import linear as LL
    from vllm_quick_amd.layers.quantization import AWQConfig
    from vllm_quick_amd.parallel_utils import parallel_state as ps
    from vllm_quick_amd.parallel_utils.communication_op import tensor_model_parallel_all_reduce
    method = AWQConfig(4, 128, True).get_linear_method()
    with ps.rehearse_one_rank(8, rank=3):
        assert ps.get_tensor_model_parallel_world_size() == 8 and ps.get_tensor_model_parallel_rank() == 3
        col = LL.ColumnParallelLinear(8192, 7168 * 8, bias=False, params_dtype=torch.float16, linear_method=method)
        row = LL.RowParallelLinear(3584 * 8, 8192, bias=False, params_dtype=torch.float16, linear_method=method)
        assert tuple(col.qweight.shape) == (8192, 7168 // 8) and tuple(col.scales.shape) == (64, 7168)
        assert tuple(row.qweight.shape) == (3584, 8192 // 8) and tuple(row.qzeros.shape) == (28, 8192 // 8)
        x = torch.arange(12.0).reshape(3, 4)
        y = tensor_model_parallel_all_reduce(x)
        assert torch.equal(y, x) and y.data_ptr() != x.data_ptr()
        with pytest.raises(ValueError):
            with ps.rehearse_one_rank(4, rank=4):
                pass
    assert ps.get_tensor_model_parallel_world_size() == 1 and not ps.rehearsing_one_rank()
    assert tensor_model_parallel_all_reduce(x) is x
