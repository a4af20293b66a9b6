"""From a checkpoint on disk to numbers on the GPU: a synthetic AutoAWQ-format checkpoint is loaded into a
gated MLP through the reference's name mapping, and the MLP (fused gate_up + activation launch, then the
down projection) is checked against the oracle computed from the checkpoint tensors themselves."""
import numpy as np
import pytest
import torch

from oracle import awq_oracle as orc

pytestmark = pytest.mark.gpu


def test_mlp_from_checkpoint(cuda, tmp_path):
    from test_checkpoint_cpu import MLP, G, H, I
    from vllm_quick_amd import weight_utils as WU
    from vllm_quick_amd.layers.activation import gate_up_silu_and_mul
    names = {"mlp.gate_proj": (H, I), "mlp.up_proj": (H, I), "mlp.down_proj": (I, H)}
    WU.write_awq_checkpoint(str(tmp_path), names, group_size=G, seed=11)
    method = WU.get_quant_config("awq", str(tmp_path)).get_linear_method()

    class Wrap(torch.nn.Module):

        def __init__(self):
            super().__init__()
            self.mlp = MLP(method)

    model = Wrap()
    WU.load_stacked_weights(model, WU.hf_model_weights_iterator(str(tmp_path)))
    model = model.to(cuda)
    ck = {k: v.numpy() for k, v in WU.hf_model_weights_iterator(str(tmp_path))}
    x = (np.random.default_rng(4).standard_normal((6, H)) * 0.5).astype(np.float16)
    with torch.inference_mode():
        act = gate_up_silu_and_mul(model.mlp.gate_up_proj, torch.from_numpy(x).to(cuda))
        y, _ = model.mlp.down_proj(act)

    def w(name):
        return ck[f"mlp.{name}.qweight"], ck[f"mlp.{name}.scales"], ck[f"mlp.{name}.qzeros"]

    g = torch.from_numpy(orc.gemm_faithful(x, *w("gate_proj")))
    u = torch.from_numpy(orc.gemm_faithful(x, *w("up_proj")))
    want_act = (torch.nn.functional.silu(g.float()).half() * u).numpy()
    got_act = act.float().cpu().numpy()
    tol = 1.1 * np.abs(u.numpy().astype(np.float64)) * (1e-3 + 1e-2 * np.abs(g.numpy().astype(np.float64))) + \
        np.abs(torch.nn.functional.silu(g.float()).numpy()) * (1e-3 + 1e-2 * np.abs(u.numpy().astype(np.float64))) + 1e-4
    assert (np.abs(got_act - want_act.astype(np.float32)) <= tol).all()
    # the down projection, on the activations the GPU produced, against the oracle on the same numbers
    want = orc.gemm_faithful(act.cpu().numpy(), *w("down_proj")).astype(np.float32)
    assert np.allclose(y.float().cpu().numpy(), want, rtol=1e-2, atol=1e-3)
