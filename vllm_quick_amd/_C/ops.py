"""`ops` submodule with the reference's AWQ operator signatures.

Mirrors `vllm._C.ops.awq_gemm` / `awq_dequantize` (csrc/pybind.cpp:52-55,
csrc/ops.h:69-82 of the reference): same positional arguments, same result
shape/dtype/device, same exception types (ValueError for shape violations, as
pybind maps std::invalid_argument; RuntimeError for dtype/device mismatches, as
`data_ptr<at::Half>()` raises).  Everything is computed by libw4a16.so on the
GPU; CPU tensors are rejected.

The functions are asynchronous (they enqueue on torch's current stream and
return) and allocate only the result tensor, through torch's caching allocator,
so they can be captured in a HIP graph once a layer's weights have been
repacked by an earlier (warm-up) call.
"""
from __future__ import annotations

import warnings
import weakref
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import torch

from .._lib import check, lib

PACK_FACTOR = 8


# --------------------------------------------------------------------------
# argument checks
# --------------------------------------------------------------------------
def _require(t: torch.Tensor, dtype: torch.dtype, name: str) -> None:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.dtype != dtype:
        want = {torch.float16: "Half", torch.int32: "Int", torch.bfloat16: "BFloat16"}[dtype]
        raise RuntimeError(f"expected scalar type {want} but found {t.dtype} for {name}")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor: the W4A16 path has no CPU implementation")


def _geometry(kernel: torch.Tensor, scaling_factors: torch.Tensor, zeros: torch.Tensor) -> Tuple[int, int, int]:
    _require(kernel, torch.int32, "kernel")
    _require(scaling_factors, torch.float16, "scaling_factors")
    _require(zeros, torch.int32, "zeros")
    if kernel.dim() != 2 or scaling_factors.dim() != 2 or zeros.dim() != 2:
        raise ValueError("kernel, scaling_factors and zeros must be 2-D")
    K, NC = kernel.shape
    N = NC * PACK_FACTOR
    groups = scaling_factors.shape[0]
    if groups == 0 or K % groups != 0:
        raise ValueError("IC is not multiple of Group size")
    G = K // groups
    if scaling_factors.shape[1] != N:
        raise ValueError(f"scaling_factors must be [{groups}, {N}], got {tuple(scaling_factors.shape)}")
    if tuple(zeros.shape) != (groups, NC):
        raise ValueError(f"zeros must be [{groups}, {NC}], got {tuple(zeros.shape)}")
    if not (kernel.device == scaling_factors.device == zeros.device):
        raise RuntimeError("kernel, scaling_factors and zeros must be on the same device")
    return K, N, G


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def stream_is_capturing() -> bool:
    """True while torch's current stream records a HIP graph."""
    return torch.cuda.is_current_stream_capturing()


class _on_device:
    """Makes `device` current for the duration of a call (OptionalCUDAGuard, gemm_kernels.cu:398)."""

    def __init__(self, device: torch.device):
        self.idx = device.index if device.index is not None else torch.cuda.current_device()
        self.prev = None

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.idx:
            self.prev = cur
            torch.cuda.set_device(self.idx)

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)


# --------------------------------------------------------------------------
# private (MFMA-ordered) weights
# --------------------------------------------------------------------------
@dataclass
class PackedAWQ:
    """A layer's weights in libw4a16's private layout (see include/w4a16.h, w4a16_repack)."""
    packed_w: torch.Tensor  # uint8 [w4a16_packed_weight_bytes]
    packed_sz: torch.Tensor  # uint8 [w4a16_packed_sz_bytes]
    K: int
    N: int
    G: int

    @property
    def device(self) -> torch.device:
        return self.packed_w.device


def awq_repack(kernel: torch.Tensor, scaling_factors: torch.Tensor, zeros: torch.Tensor) -> PackedAWQ:
    """Stock AWQ tensors -> PackedAWQ (one-time, on the GPU; inputs are left untouched)."""
    K, N, G = _geometry(kernel, scaling_factors, zeros)
    kernel, scaling_factors, zeros = kernel.contiguous(), scaling_factors.contiguous(), zeros.contiguous()
    wbytes = lib.w4a16_packed_weight_bytes(K, N)
    sbytes = lib.w4a16_packed_sz_bytes(K, N, G)
    if wbytes == 0 or sbytes == 0:
        raise ValueError(f"cannot repack K={K} N={N} G={G}")
    with _on_device(kernel.device):
        pw = torch.empty(wbytes, dtype=torch.uint8, device=kernel.device)
        ps = torch.empty(sbytes, dtype=torch.uint8, device=kernel.device)
        check(
            lib.w4a16_repack(kernel.data_ptr(), scaling_factors.data_ptr(), zeros.data_ptr(), pw.data_ptr(),
                             ps.data_ptr(), K, N, G, _stream()), "awq_repack")
    return PackedAWQ(pw, ps, K, N, G)


def gemm_workspace(device: torch.device, nbytes: int = 4 << 20) -> torch.Tensor:
    """A zeroed device buffer for launches that split K over workgroups (partial sums and ticket counters).

    Zero it once: the kernels return every counter they used to zero.  One workspace serves any number of
    layers, but only one launch at a time (launches on different streams need one each)."""
    return torch.zeros(max(int(nbytes), 16), dtype=torch.uint8, device=device)


_STREAM_WORKSPACES: dict = {}


def stream_workspace(device: torch.device) -> Optional[torch.Tensor]:
    """The workspace of torch's current stream on `device` (made on first use, 1 MiB): launches on one stream
    follow each other, so they can share one; another stream gets its own.  None while a graph is being captured
    and the stream has none yet (allocating and zeroing inside a capture is not wanted): the call then runs
    without a split."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream)
    ws = _STREAM_WORKSPACES.get(key)
    if ws is None:
        if stream_is_capturing():
            return None
        ws = _STREAM_WORKSPACES[key] = gemm_workspace(torch.device("cuda", idx), 1 << 20)
    return ws


def _check_workspace(workspace: torch.Tensor, device: torch.device) -> None:
    if not workspace.is_cuda or workspace.device != device or not workspace.is_contiguous():
        raise RuntimeError("workspace must be a contiguous tensor on the device of the activations")


def awq_gemm_packed(in_feats: torch.Tensor,
                    packed: PackedAWQ,
                    bias: Optional[torch.Tensor] = None,
                    cfg: Optional[Tuple[int, int, int]] = None,
                    workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[M, K] fp16 @ packed weights (+ bias) -> [M, N] fp16.  `cfg` = (mt, t, waves), (mt, t, nw, kw) for the LDS-staged kernel, ("wide", mt, t, kw) for the 32x32x16 kernel or ("decode", t, waves, splits) for the decode kernel overrides the launch shape.  `workspace` (see `gemm_workspace`) lets the library split K over workgroups where that pays."""
    if isinstance(in_feats, torch.Tensor) and in_feats.dtype == torch.bfloat16:
        return awq_gemm_packed_bf16(in_feats, packed, bias, cfg)  # (no workspace: the bf16 kernel splits nothing)
    _require(in_feats, torch.float16, "in_feats")
    if in_feats.dim() != 2 or in_feats.shape[1] != packed.K:
        raise ValueError(f"in_feats must be [M, {packed.K}], got {tuple(in_feats.shape)}")
    if in_feats.device != packed.device:
        raise RuntimeError("in_feats and the packed weights are on different devices")
    x = in_feats.contiguous()  # the reference reads raw data_ptr and assumes this (gemm_kernels.cu:405)
    M = x.shape[0]
    bias_ptr = None
    if bias is not None:
        _require(bias, torch.float16, "bias")
        if bias.numel() != packed.N:
            raise ValueError(f"bias must have {packed.N} elements")
        bias = bias.contiguous()
        bias_ptr = bias.data_ptr()
    with _on_device(x.device):
        out = torch.empty((M, packed.N), dtype=torch.float16, device=x.device)
        if M == 0:
            return out
        if cfg is None and workspace is not None:
            _check_workspace(workspace, x.device)
            rc = lib.w4a16_gemm_packed_f16_ws(x.data_ptr(), 0, packed.packed_w.data_ptr(), packed.packed_sz.data_ptr(),
                                              bias_ptr, out.data_ptr(), 0, M, packed.K, packed.N, packed.G, 1, 0,
                                              workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                              _stream())
        elif cfg is None:
            rc = lib.w4a16_gemm_packed_f16(x.data_ptr(), packed.packed_w.data_ptr(), packed.packed_sz.data_ptr(),
                                           bias_ptr, out.data_ptr(), M, packed.K, packed.N, packed.G, _stream())
        elif cfg[0] == "decode":  # ("decode", t, waves, splits): up to 4 rows
            _, t, waves, splits = cfg[:4]
            depth = cfg[4] if len(cfg) > 4 else 0  # blocks a wave keeps in flight (0: the library's default)
            if splits > 1 and workspace is None:
                workspace = gemm_workspace(x.device, lib.w4a16_decode_workspace_bytes(packed.K, packed.N, 1, 0, t, waves, splits))
            ws_ptr, ws_bytes = (None, 0)
            if workspace is not None:
                _check_workspace(workspace, x.device)
                ws_ptr, ws_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
            rc = lib.w4a16_gemm_packed_f16_decode_cfg(x.data_ptr(), 0, packed.packed_w.data_ptr(),
                                                      packed.packed_sz.data_ptr(), bias_ptr, out.data_ptr(), 0, M,
                                                      packed.K, packed.N, packed.G, 1, 0, t, waves, depth, splits, ws_ptr,
                                                      ws_bytes, _stream())
        elif cfg[0] == "wide" and len(cfg) > 4 and cfg[4] > 1:  # ("wide", mt, t, kw, splits): K split over workgroups
            _, mt, t, kw, splits = cfg
            if workspace is None:
                workspace = gemm_workspace(x.device, lib.w4a16_wide_workspace_bytes(M, packed.N, 1, 0, mt, t, splits))
            _check_workspace(workspace, x.device)
            rc = lib.w4a16_gemm_packed_f16_wide_split_cfg(x.data_ptr(), packed.packed_w.data_ptr(),
                                                          packed.packed_sz.data_ptr(), bias_ptr, out.data_ptr(), M,
                                                          packed.K, packed.N, packed.G, 0, mt, t, kw, splits,
                                                          workspace.data_ptr(),
                                                          workspace.numel() * workspace.element_size(), _stream())
        elif cfg[0] == "wide":  # the 32x32x16 kernel: ("wide", mt, t, kw), tiles of 32 rows / 32 columns
            _, mt, t, kw = cfg[:4]
            rc = lib.w4a16_gemm_packed_f16_wide_cfg(x.data_ptr(), packed.packed_w.data_ptr(),
                                                    packed.packed_sz.data_ptr(), bias_ptr, out.data_ptr(), M,
                                                    packed.K, packed.N, packed.G, mt, t, kw, _stream())
        elif len(cfg) == 4:
            mt, t, nw, kw = cfg
            rc = lib.w4a16_gemm_packed_f16_tiled_cfg(x.data_ptr(), packed.packed_w.data_ptr(),
                                                     packed.packed_sz.data_ptr(), bias_ptr, out.data_ptr(), M,
                                                     packed.K, packed.N, packed.G, mt, t, nw, kw, _stream())
        else:
            mt, t, waves = cfg
            rc = lib.w4a16_gemm_packed_f16_cfg(x.data_ptr(), packed.packed_w.data_ptr(),
                                               packed.packed_sz.data_ptr(), bias_ptr, out.data_ptr(), M, packed.K,
                                               packed.N, packed.G, mt, t, waves, _stream())
        check(rc, "awq_gemm")
    return out


def awq_gemm_packed_bf16(in_feats: torch.Tensor,
                         packed: PackedAWQ,
                         bias: Optional[torch.Tensor] = None,
                         cfg: Optional[Tuple[int, int, int]] = None,
                         out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[M, K] bf16 @ packed weights (+ bias, bf16) -> [M, N] bf16, with weights bf16((q - z) * s) rounded once and
    fp32 accumulation.  An extension: the reference's AWQ path takes fp16 only (awq.py:42-43), so nothing in it pins
    these results; the scales stay fp16 as the checkpoint holds them.  `cfg` = a launch shape as
    `w4a16_built_shape_bf16` lists them: (mt, t, waves) of the streaming kernel, (-16, t, waves) of the decode kernel
    (up to 16 rows), (-mt, t, kw) of the 32x32x16 kernel; `out` ([M, N] bf16, contiguous): the result is ADDED to it in place (bf16 + bf16, rounded
    once) and returned."""
    _require(in_feats, torch.bfloat16, "in_feats")
    if in_feats.dim() != 2 or in_feats.shape[1] != packed.K:
        raise ValueError(f"in_feats must be [M, {packed.K}], got {tuple(in_feats.shape)}")
    if in_feats.device != packed.device:
        raise RuntimeError("in_feats and the packed weights are on different devices")
    x = in_feats.contiguous()
    M = x.shape[0]
    bias_ptr = None
    if bias is not None:
        _require(bias, torch.bfloat16, "bias")
        if bias.numel() != packed.N:
            raise ValueError(f"bias must have {packed.N} elements")
        bias = bias.contiguous()
        bias_ptr = bias.data_ptr()
    epilogue = 0
    if out is not None:
        _require(out, torch.bfloat16, "out")
        if tuple(out.shape) != (M, packed.N) or not out.is_contiguous() or out.device != x.device:
            raise ValueError(f"out must be a contiguous [{M}, {packed.N}] tensor on the device of in_feats")
        epilogue = 2  # W4A16_EPI_ADD_TO_OUT
    mt, t, waves = cfg if cfg is not None else (0, 0, 0)
    with _on_device(x.device):
        if out is None:
            out = torch.empty((M, packed.N), dtype=torch.bfloat16, device=x.device)
        if M == 0:
            return out
        rc = lib.w4a16_gemm_packed_bf16(x.data_ptr(), packed.packed_w.data_ptr(), packed.packed_sz.data_ptr(), bias_ptr,
                                        out.data_ptr(), M, packed.K, packed.N, packed.G, epilogue, mt, t, waves, _stream())
        check(rc, "awq_gemm (bf16)")
    return out


@dataclass
class PackedAWQBatch:
    """E same-shaped weight matrices in the private layout, back to back (the form
    w4a16_gemm_packed_f16_batched reads): e.g. the experts of a quantised Mixtral block."""
    packed_w: torch.Tensor  # uint8 [E, w4a16_packed_weight_bytes]
    packed_sz: torch.Tensor  # uint8 [E, w4a16_packed_sz_bytes]
    K: int
    N: int
    G: int

    @property
    def device(self) -> torch.device:
        return self.packed_w.device

    @property
    def batch(self) -> int:
        return self.packed_w.shape[0]

    def __getitem__(self, e: int) -> PackedAWQ:
        return PackedAWQ(self.packed_w[e], self.packed_sz[e], self.K, self.N, self.G)


def awq_repack_batch(kernels, scaling_factors, zeros) -> PackedAWQBatch:
    """Sequences (or stacked [E, ...] tensors) of same-shaped stock AWQ tensors -> PackedAWQBatch."""
    E = len(kernels)
    if E == 0 or len(scaling_factors) != E or len(zeros) != E:
        raise ValueError("need the same, non-zero number of kernels, scaling_factors and zeros")
    K, N, G = _geometry(kernels[0], scaling_factors[0], zeros[0])
    wbytes = lib.w4a16_packed_weight_bytes(K, N)
    sbytes = lib.w4a16_packed_sz_bytes(K, N, G)
    if wbytes == 0 or sbytes == 0:
        raise ValueError(f"cannot repack K={K} N={N} G={G}")
    dev = kernels[0].device
    with _on_device(dev):
        pw = torch.empty((E, wbytes), dtype=torch.uint8, device=dev)
        ps = torch.empty((E, sbytes), dtype=torch.uint8, device=dev)
        for e in range(E):
            if _geometry(kernels[e], scaling_factors[e], zeros[e]) != (K, N, G) or kernels[e].device != dev:
                raise ValueError("all problems of a batch must have one shape and one device")
            k, s, z = kernels[e].contiguous(), scaling_factors[e].contiguous(), zeros[e].contiguous()
            check(
                lib.w4a16_repack(k.data_ptr(), s.data_ptr(), z.data_ptr(), pw[e].data_ptr(), ps[e].data_ptr(), K, N,
                                 G, _stream()), "awq_repack")
    return PackedAWQBatch(pw, ps, K, N, G)


def _problem_mask(mask: Optional[torch.Tensor], E: int, device: torch.device, cfg) -> Optional[torch.Tensor]:
    if mask is None:
        return None
    if cfg is not None:
        raise ValueError("a problem mask goes with the library's own launch shape (cfg=None)")
    if mask.dtype != torch.int32 or tuple(mask.shape) != (E, ) or mask.device != device or not mask.is_contiguous():
        raise ValueError(f"mask must be a contiguous int32 [{E}] tensor on {device}")
    return mask


def awq_gemm_packed_batched(in_feats: torch.Tensor,
                            packed: PackedAWQBatch,
                            bias: Optional[torch.Tensor] = None,
                            cfg: Optional[Tuple[int, ...]] = None,
                            mask: Optional[torch.Tensor] = None,
                            active_hint: int = 0,
                            zero_masked: bool = True) -> torch.Tensor:
    """One launch for E GEMMs of one shape: in_feats [M, K] (shared by all problems) or [E, M, K] fp16,
    optional bias [E, N] -> [E, M, N] fp16.  Stands in for the reference's loop over experts
    (mixtral_quant.py:145-157).  `mask` (int32 [E] on the device, read by the kernel): problems whose entry
    is 0 are skipped, their weights never read, and their rows of the result are zero (`zero_masked=False`:
    left as allocated, for a caller that never looks at them, one fill kernel less); `active_hint` (how
    many the caller expects to be on, 0 = unknown) steers the launch shape only."""
    _require(in_feats, torch.float16, "in_feats")
    E = packed.batch
    if in_feats.dim() == 2:
        shared = True
    elif in_feats.dim() == 3 and in_feats.shape[0] == E:
        shared = False
    else:
        raise ValueError(f"in_feats must be [M, {packed.K}] or [{E}, M, {packed.K}], got {tuple(in_feats.shape)}")
    if in_feats.shape[-1] != packed.K:
        raise ValueError(f"in_feats must be [..., {packed.K}], got {tuple(in_feats.shape)}")
    if in_feats.device != packed.device:
        raise RuntimeError("in_feats and the packed weights are on different devices")
    x = in_feats.contiguous()
    M = x.shape[-2]
    bias_ptr = None
    if bias is not None:
        _require(bias, torch.float16, "bias")
        if tuple(bias.shape) != (E, packed.N):
            raise ValueError(f"bias must be [{E}, {packed.N}]")
        bias = bias.contiguous()
        bias_ptr = bias.data_ptr()
    mask = _problem_mask(mask, E, x.device, cfg)
    with _on_device(x.device):
        out = (torch.zeros if mask is not None and zero_masked else torch.empty)((E, M, packed.N), dtype=torch.float16,
                                                                                 device=x.device)
        if M == 0:
            return out
        if (M * packed.K) % 8 and not shared:
            raise ValueError("M * K must be a multiple of 8 for per-problem activations")
        if (M * packed.N) % 4:
            raise ValueError("M * N must be a multiple of 4")
        common = (x.data_ptr(), 0 if shared else M * packed.K, packed.packed_w.data_ptr(),
                  packed.packed_sz.data_ptr(), bias_ptr, out.data_ptr(), M * packed.N, M, packed.K, packed.N,
                  packed.G, E)
        if mask is not None:
            rc = lib.w4a16_gemm_packed_f16_masked(*common, EPI_NONE, mask.data_ptr(), int(active_hint), _stream())
        elif cfg is None:
            rc = lib.w4a16_gemm_packed_f16_batched(*common, _stream())
        else:  # (mt, t, waves) = streaming kernel, (mt, t, nw, kw) = LDS-staged kernel
            mt, t, a = cfg[:3]
            rc = lib.w4a16_gemm_packed_f16_batched_cfg(*common, mt, t, a, cfg[3] if len(cfg) == 4 else 0, _stream())
        check(rc, "awq_gemm_batched")
    return out


EPI_NONE, EPI_SILU_MUL, EPI_ADD_TO_OUT = 0, 1, 2


def awq_gemm_packed_silu_and_mul(in_feats: torch.Tensor,
                                 packed,
                                 bias: Optional[torch.Tensor] = None,
                                 cfg: Optional[Tuple[int, ...]] = None,
                                 mask: Optional[torch.Tensor] = None,
                                 active_hint: int = 0,
                                 zero_masked: bool = True) -> torch.Tensor:
    """silu(x @ W[:, :N/2]) * (x @ W[:, N/2:]) in one launch: the merged gate_up projection of a gated
    MLP with its activation fused into the GEMM epilogue (llama.py:74-80 of the reference runs the GEMM,
    then `silu_and_mul`).  `packed` is a PackedAWQ ([M, K] -> [M, N/2]) or a PackedAWQBatch ([M, K] shared
    or [E, M, K] -> [E, M, N/2]); bias, if given, covers all N columns.  `mask`: as for
    `awq_gemm_packed_batched` (batches only)."""
    _require(in_feats, torch.float16, "in_feats")
    batched = isinstance(packed, PackedAWQBatch)
    E = packed.batch if batched else 1
    if packed.N % 32:
        raise ValueError("the fused activation needs two halves of whole 16-column tiles (N % 32 == 0)")
    if in_feats.shape[-1] != packed.K or in_feats.dim() not in ((2, 3) if batched else (2, )):
        raise ValueError(f"in_feats must be [M, {packed.K}]" + (f" or [{E}, M, {packed.K}]" if batched else "") +
                         f", got {tuple(in_feats.shape)}")
    shared = in_feats.dim() == 2
    if not shared and in_feats.shape[0] != E:
        raise ValueError(f"in_feats must carry {E} problems, got {in_feats.shape[0]}")
    if in_feats.device != packed.device:
        raise RuntimeError("in_feats and the packed weights are on different devices")
    x = in_feats.contiguous()
    M, half = x.shape[-2], packed.N // 2
    bias_ptr = None
    if bias is not None:
        _require(bias, torch.float16, "bias")
        if bias.numel() != E * packed.N:
            raise ValueError(f"bias must have {E * packed.N} elements")
        bias = bias.contiguous()
        bias_ptr = bias.data_ptr()
    if mask is not None and not batched:
        raise ValueError("a problem mask needs a PackedAWQBatch")
    mask = _problem_mask(mask, E, x.device, cfg)
    with _on_device(x.device):
        out = (torch.zeros if mask is not None and zero_masked else torch.empty)(
            (E, M, half) if batched else (M, half), dtype=torch.float16, device=x.device)
        if M == 0:
            return out
        if batched and ((M * half) % 4 or (not shared and (M * packed.K) % 8)):
            raise ValueError("per-problem buffers must stay 8-byte (out) / 16-byte (in_feats) aligned")
        mt, t, a, b = (0, 0, 0, 0) if cfg is None else (tuple(cfg) + (0, ))[:4]
        head = (x.data_ptr(), 0 if shared else M * packed.K, packed.packed_w.data_ptr(), packed.packed_sz.data_ptr(),
                bias_ptr, out.data_ptr(), M * half, M, packed.K, packed.N, packed.G, E, EPI_SILU_MUL)
        if mask is not None:
            rc = lib.w4a16_gemm_packed_f16_masked(*head, mask.data_ptr(), int(active_hint), _stream())
        else:
            rc = lib.w4a16_gemm_packed_f16_ex(*head, mt, t, a, b, _stream())
        check(rc, "awq_gemm_silu_and_mul")
    return out


def moe_route(router_logits: torch.Tensor, top_k: int, first_local: int = 0, n_local: Optional[int] = None,
              want_mask: bool = False) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Routing weights of a Mixtral-style block in one launch (mixtral_quant.py:137-150: softmax over all
    experts in fp32, top-k, renormalise, scatter): router_logits [T, E] fp16 -> weights [T, n_local] fp32 for
    the experts [first_local, first_local + n_local) (0 where a token did not select the expert) and, with
    `want_mask`, the int32 [n_local] problem mask of the batched launches (1 = some token selected it)."""
    _require(router_logits, torch.float16, "router_logits")
    if router_logits.dim() != 2:
        raise ValueError(f"router_logits must be [tokens, experts], got {tuple(router_logits.shape)}")
    T, E = router_logits.shape
    n_local = E - first_local if n_local is None else n_local
    logits = router_logits.contiguous()
    with _on_device(logits.device):
        weights = torch.empty((T, n_local), dtype=torch.float32, device=logits.device)
        mask = torch.empty((n_local, ), dtype=torch.int32, device=logits.device) if want_mask else None
        if T == 0:
            return weights, None if mask is None else mask.zero_()
        check(lib.w4a16_moe_route_f16(logits.data_ptr(), weights.data_ptr(), None if mask is None else mask.data_ptr(),
                                      T, E, int(top_k), int(first_local), int(n_local), _stream()), "moe_route")
    return weights, mask


def moe_combine(y: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    """out[t] = sum over experts e of fp16(y[e, t] * weights[t, e]), summed in fp32 and rounded once
    (mixtral_quant.py:152-157), in one launch: y [E, T, H] fp16 (a batched launch's output), weights [T, E]
    fp32 as `moe_route` returns them.  Experts with weight 0 for a token are not read."""
    _require(y, torch.float16, "y")
    _require(weights, torch.float32, "weights")
    if y.dim() != 3 or tuple(weights.shape) != (y.shape[1], y.shape[0]):
        raise ValueError(f"y must be [E, T, H] and weights [T, E], got {tuple(y.shape)} and {tuple(weights.shape)}")
    if y.device != weights.device:
        raise RuntimeError("y and weights are on different devices")
    E, T, H = y.shape
    y, weights = y.contiguous(), weights.contiguous()
    with _on_device(y.device):
        out = torch.empty((T, H), dtype=torch.float16, device=y.device)
        if T == 0:
            return out
        check(lib.w4a16_moe_combine_f16(y.data_ptr(), weights.data_ptr(), out.data_ptr(), E, T, H, _stream()),
              "moe_combine")
    return out


def awq_gemm_packed_add_(out: torch.Tensor,
                         in_feats: torch.Tensor,
                         packed: PackedAWQ,
                         bias: Optional[torch.Tensor] = None,
                         cfg: Optional[Tuple[int, ...]] = None) -> torch.Tensor:
    """out += in_feats @ W (+ bias), in place, in one launch: `residual + linear(x)` of a decoder layer
    with `out` the residual stream.  The GEMM result is rounded to fp16 (bias added in fp16) and then
    added in fp16, exactly as the two separate operations would."""
    _require(in_feats, torch.float16, "in_feats")
    _require(out, torch.float16, "out")
    if in_feats.dim() != 2 or in_feats.shape[1] != packed.K:
        raise ValueError(f"in_feats must be [M, {packed.K}], got {tuple(in_feats.shape)}")
    M = in_feats.shape[0]
    if tuple(out.shape) != (M, packed.N) or not out.is_contiguous():
        raise ValueError(f"out must be a contiguous [{M}, {packed.N}] tensor, got {tuple(out.shape)}")
    if in_feats.device != packed.device or out.device != packed.device:
        raise RuntimeError("in_feats, out and the packed weights must be on one device")
    x = in_feats.contiguous()
    bias_ptr = None
    if bias is not None:
        _require(bias, torch.float16, "bias")
        if bias.numel() != packed.N:
            raise ValueError(f"bias must have {packed.N} elements")
        bias = bias.contiguous()
        bias_ptr = bias.data_ptr()
    if M == 0:
        return out
    with _on_device(x.device):
        mt, t, a, b = (0, 0, 0, 0) if cfg is None else (tuple(cfg) + (0, ))[:4]
        rc = lib.w4a16_gemm_packed_f16_ex(x.data_ptr(), 0, packed.packed_w.data_ptr(), packed.packed_sz.data_ptr(),
                                          bias_ptr, out.data_ptr(), M * packed.N, M, packed.K, packed.N, packed.G, 1,
                                          EPI_ADD_TO_OUT, mt, t, a, b, _stream())
        check(rc, "awq_gemm_add_")
    return out


def awq_gemm_stock(in_feats: torch.Tensor, kernel: torch.Tensor, scaling_factors: torch.Tensor,
                   zeros: torch.Tensor) -> torch.Tensor:
    """Shape-agnostic GEMM straight from the stock tensors (no repack, nothing cached)."""
    K, N, G = _geometry(kernel, scaling_factors, zeros)
    _require(in_feats, torch.float16, "in_feats")
    if in_feats.dim() != 2 or in_feats.shape[1] != K:
        raise ValueError(f"in_feats must be [M, {K}], got {tuple(in_feats.shape)}")
    x = in_feats.contiguous()
    kernel, scaling_factors, zeros = kernel.contiguous(), scaling_factors.contiguous(), zeros.contiguous()
    M = x.shape[0]
    with _on_device(x.device):
        out = torch.empty((M, N), dtype=torch.float16, device=x.device)
        if M:
            check(
                lib.w4a16_gemm_f16(x.data_ptr(), kernel.data_ptr(), scaling_factors.data_ptr(), zeros.data_ptr(),
                                   out.data_ptr(), M, K, N, G, PACK_FACTOR, _stream()), "awq_gemm")
    return out


# Repack cache for callers that hand over stock tensors on every call (the
# reference signature).  An entry is reused only for the very same tensor
# objects at the same version counters, so in-place writes PyTorch counts,
# freed and re-used addresses and fresh views all miss and re-pack.  Two kinds
# of write are invisible to a version counter: writes through `param.data`
# (the reference's loaders: call clear_repack_cache(), or mark_written() on this
# package's layers, after such a load that follows a first call), and writes to
# tensors created under torch.inference_mode(), which keep no counter at all:
# nothing is cached for those (every call re-packs, warned about once).
def tensor_version(t: torch.Tensor) -> int:
    """In-place-write counter of a tensor; inference-mode tensors keep none and report 0 (`_cached_pack` never
    keeps an entry for them)."""
    try:
        return t._version
    except RuntimeError:
        return 0


class _CacheEntry:
    __slots__ = ("refs", "versions", "packed")

    def __init__(self, tensors, packed):
        self.refs = tuple(weakref.ref(t) for t in tensors)
        self.versions = tuple(tensor_version(t) for t in tensors)
        self.packed = packed

    def matches(self, tensors) -> bool:
        return all(r() is t for r, t in zip(self.refs, tensors)) and \
            self.versions == tuple(tensor_version(t) for t in tensors)


_repack_cache: Dict[Tuple, _CacheEntry] = {}
_CACHE_LIMIT = 4096
_warned_untracked = False


def _cached_pack(kernel, scaling_factors, zeros) -> Optional[PackedAWQ]:
    key = (kernel.data_ptr(), scaling_factors.data_ptr(), zeros.data_ptr(), tuple(kernel.shape),
           scaling_factors.shape[0], kernel.device.index)
    tensors = (kernel, scaling_factors, zeros)
    if any(t.is_inference() for t in tensors):
        # no version counters: an in-place reload could not be told from no change, so nothing is kept
        if stream_is_capturing():
            return None
        global _warned_untracked
        if not _warned_untracked:
            _warned_untracked = True
            warnings.warn("awq_gemm: the weights are inference-mode tensors, which keep no version counter: they are "
                          "re-ordered into the private layout on every call; create weights outside "
                          "torch.inference_mode() to have that done once", RuntimeWarning, stacklevel=3)
        return awq_repack(kernel, scaling_factors, zeros)
    entry = _repack_cache.get(key)
    if entry is not None and entry.matches(tensors):
        return entry.packed
    if stream_is_capturing():
        return None  # never re-layout or allocate persistent state inside a graph capture
    if len(_repack_cache) >= _CACHE_LIMIT:
        dead = [k for k, e in _repack_cache.items() if any(r() is None for r in e.refs)]
        for k in dead:
            del _repack_cache[k]
        if len(_repack_cache) >= _CACHE_LIMIT:
            _repack_cache.clear()
    packed = awq_repack(kernel, scaling_factors, zeros)
    # once per layer: the copy is complete before a call on any other stream can find it in the cache
    torch.cuda.current_stream(kernel.device).synchronize()
    _repack_cache[key] = _CacheEntry(tensors, packed)
    return packed


def clear_repack_cache() -> None:
    _repack_cache.clear()


# --------------------------------------------------------------------------
# the reference's two operators
# --------------------------------------------------------------------------
def awq_gemm(_in_feats: torch.Tensor, _kernel: torch.Tensor, _scaling_factors: torch.Tensor, _zeros: torch.Tensor,
             split_k_iters: int) -> torch.Tensor:
    """Reference: csrc/quantization/awq/gemm_kernels.cu:389-446.

    [M, IC] fp16, [IC, OC/8] int32, [IC/G, OC] fp16, [IC/G, OC/8] int32 -> new [M, OC] fp16.
    `split_k_iters` is accepted and ignored.  OC % 64 is not required (the
    reference needs it, gemm_kernels.cu:412-413; 1376-wide TP shards do not
    satisfy it); OC % 8, G % 32 and IC % G are.
    """
    K, N, G = _geometry(_kernel, _scaling_factors, _zeros)
    # bf16 activations (an extension; the reference takes fp16 only) run on the packed copy alone
    bf16 = isinstance(_in_feats, torch.Tensor) and _in_feats.dtype == torch.bfloat16
    _require(_in_feats, torch.bfloat16 if bf16 else torch.float16, "in_feats")
    if _in_feats.dim() != 2 or _in_feats.shape[1] != K:
        raise ValueError(f"in_feats must be [M, {K}], got {tuple(_in_feats.shape)}")
    if G % 32 != 0:
        raise ValueError("Group size should be a multiple of 32")
    packed = None
    if _kernel.is_contiguous() and _scaling_factors.is_contiguous() and _zeros.is_contiguous():
        packed = _cached_pack(_kernel, _scaling_factors, _zeros)
    if packed is None:
        # No packed copy can be made now (non-contiguous tensors, or the stream is being captured and this
        # layer was never called before): the stock-layout kernel serves the call, three to five times slower,
        # and if this is a capture the graph keeps it.  Counted, and said once.
        if bf16:
            raise RuntimeError("awq_gemm: bf16 activations need the packed copy of the weights, which cannot be made "
                               "now (non-contiguous weight tensors, or a first call on these weights while a HIP graph "
                               "is being captured): run one eager call per layer first")
        global _stock_calls
        _stock_calls += 1
        if _stock_calls == 1:
            warnings.warn("awq_gemm: a call was served by the stock-layout kernel (first call on these weights "
                          "while a HIP graph is being captured, or non-contiguous weights); run one eager call per "
                          "layer before capturing to get the packed-layout kernels", RuntimeWarning, stacklevel=2)
        return awq_gemm_stock(_in_feats, _kernel, _scaling_factors, _zeros)
    return awq_gemm_packed(_in_feats, packed)


_stock_calls = 0


def stock_kernel_calls() -> int:
    """How many awq_gemm calls the stock-layout kernel has served in this process."""
    return _stock_calls


def awq_dequantize(_kernel: torch.Tensor, _scaling_factors: torch.Tensor, _zeros: torch.Tensor, split_k_iters: int,
                   thx: int, thy: int) -> torch.Tensor:
    """Reference: csrc/quantization/awq/gemm_kernels.cu:332-381.

    -> new [IC, OC] tensor with the dtype/device of `_scaling_factors`.
    `split_k_iters`, `thx`, `thy` only shaped the reference's launch; ignored.
    """
    K, N, G = _geometry(_kernel, _scaling_factors, _zeros)
    if G % 32 != 0:
        raise ValueError("Group size should be a multiple of 32")
    kernel, scaling_factors, zeros = _kernel.contiguous(), _scaling_factors.contiguous(), _zeros.contiguous()
    with _on_device(kernel.device):
        out = torch.empty((K, N), dtype=torch.float16, device=kernel.device)
        check(
            lib.w4a16_dequantize_f16(kernel.data_ptr(), scaling_factors.data_ptr(), zeros.data_ptr(), out.data_ptr(),
                                     K, N, G, _stream()), "awq_dequantize")
    return out


def awq_dequantize_packed(packed: PackedAWQ) -> torch.Tensor:
    """[IC, OC] fp16 from the private layout (bit-identical to awq_dequantize on the source tensors)."""
    with _on_device(packed.device):
        out = torch.empty((packed.K, packed.N), dtype=torch.float16, device=packed.device)
        check(
            lib.w4a16_dequantize_packed_f16(packed.packed_w.data_ptr(), packed.packed_sz.data_ptr(), out.data_ptr(),
                                            packed.K, packed.N, packed.G, _stream()), "awq_dequantize_packed")
    return out
