"""Oracle-backed stand-ins for the native operators, for tests that run without a GPU.
They live under tests/ because only tests may compute with the oracle."""
import torch

from oracle import awq_oracle as orc


class FakePacked(tuple):
    """Stands in for ops.PackedAWQ: an independent copy of the stock tensors."""

    @property
    def N(self):
        return self[0].shape[1] * 8

    @property
    def K(self):
        return self[0].shape[0]


class FakeOps:

    def __init__(self):
        self.trace = []

    def stream_is_capturing(self):
        return False

    @staticmethod
    def tensor_version(t):
        return t._version

    def awq_repack(self, qweight, scales, qzeros):
        self.trace.append(["repack"])
        return FakePacked((qweight.detach().clone(), scales.detach().clone(), qzeros.detach().clone()))

    def awq_dequantize_packed(self, packed):
        self.trace.append(["awq_dequantize_packed"])
        qw, s, qz = packed
        return torch.from_numpy(orc.dequantize(qw.numpy(), s.numpy(), qz.numpy()))

    def awq_gemm_packed(self, x, packed, bias=None, cfg=None):
        self.trace.append(["awq_gemm", list(x.shape), bool(x.is_contiguous())])
        qw, s, qz = packed
        y = torch.from_numpy(
            orc.gemm_faithful(x.numpy(), qw.detach().numpy(), s.detach().numpy(), qz.detach().numpy()))
        return y if bias is None else y + bias

    def awq_dequantize(self, qweight, scales, qzeros, split_k, thx, thy):
        self.trace.append(["awq_dequantize", list(qweight.shape), split_k, thx, thy])
        return torch.from_numpy(
            orc.dequantize(qweight.detach().numpy(), scales.detach().numpy(), qzeros.detach().numpy()))

    def awq_repack_batch(self, kernels, scaling_factors, zeros):
        self.trace.append(["repack_batch", len(kernels)])
        return list(zip(kernels, scaling_factors, zeros))

    def awq_gemm_packed_batched(self, x, packed, bias=None, mask=None, active_hint=0):
        self.trace.append(["awq_gemm_batched", list(x.shape), len(packed)] +
                          ([] if mask is None else [mask.tolist()]))
        outs = []
        for e, (qw, s, qz) in enumerate(packed):
            xe = x if x.dim() == 2 else x[e]
            y = torch.from_numpy(
                orc.gemm_faithful(xe.contiguous().numpy(), qw.detach().numpy(), s.detach().numpy(),
                                  qz.detach().numpy()))
            y = y if bias is None else y + bias[e]
            outs.append(y if mask is None or int(mask[e]) else torch.zeros_like(y))  # skipped problems read as zero
        return torch.stack(outs)

    def awq_gemm_packed_silu_and_mul(self, x, packed, bias=None, cfg=None, mask=None, active_hint=0):
        self.trace.append(["awq_gemm_silu_and_mul", list(x.shape)] + ([] if mask is None else [mask.tolist()]))
        if isinstance(packed, list):  # a batch: list of (qweight, scales, qzeros)
            y = self.awq_gemm_packed_batched(x, packed, bias, mask=mask)
        else:
            y = self.awq_gemm_packed(x, packed, bias)
        self.trace.pop()
        half = y.shape[-1] // 2
        g = y[..., :half].float()
        return (g / (1.0 + torch.exp(-g))).half() * y[..., half:]

    def awq_gemm_packed_add_(self, out, x, packed, bias=None, cfg=None):
        self.trace.append(["awq_gemm_add_", list(x.shape)])
        y = self.awq_gemm_packed(x, packed, bias)
        self.trace.pop()
        return out.add_(y)
