"""Model-level measurement harness: greedy decode of a Llama-2-shaped AWQ model whose linear layers
are this package's drop-in layers (`vllm_quick_amd.layers.linear` + `AWQLinearMethod`).

This is a measurement stand-in for the reference's engine (scheduler, paged attention, sampler are out
of scope): the non-GEMM pieces are stock PyTorch ops over a contiguous KV cache (RMSNorm, RoPE,
`scaled_dot_product_attention`, SiLU-and-mul, fp16 lm_head + argmax), weights are synthetic and
seeded (no checkpoint can be fetched), and one decode step per batch size is captured in a HIP graph
the way the reference captures its decode batches (vllm/worker/model_runner.py:658-737).
Layer order follows vllm/model_executor/models/llama.py:53-80 (MLP), :119-159 (attention), :197-223
(decoder layer).  tokens/s = batch size / median step time (decode-only variant of
benchmarks/benchmark_latency.py:59-65).

    python benchmarks/llama_decode.py [--model 7b|70b] [--bs 1,2,4,8,16,32,64] [--layers N]
    torchrun --nproc-per-node 8 benchmarks/llama_decode.py --model 70b      # tensor parallel
    torchrun --nproc-per-node 2 benchmarks/llama_decode.py --backend gloo --same-device --p2p-all-reduce
                                                                            # rehearsal on one GPU
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from vllm_quick_amd._C import ops  # noqa: E402,F401
from vllm_quick_amd.layers.activation import gate_up_silu_and_mul  # noqa: E402
from vllm_quick_amd.layers.linear import (MergedColumnParallelLinear, QKVParallelLinear,  # noqa: E402
                                          RowParallelLinear, row_parallel_add_)
from vllm_quick_amd.layers.quantization import AWQConfig  # noqa: E402
from vllm_quick_amd.parallel_utils import parallel_state as ps  # noqa: E402

SHAPES = {
    # hidden, intermediate, heads, kv_heads, layers, vocab
    "7b": dict(hidden=4096, inter=11008, heads=32, kv_heads=32, layers=32, vocab=32000),
    "70b": dict(hidden=8192, inter=28672, heads=64, kv_heads=8, layers=80, vocab=32000),
}


def fill_awq(layer, gen, device):
    """Seeded synthetic AWQ tensors: uniform bit patterns, scales U(1e-3, 2e-2) (the reference's
    dummy loader leaves the int tensors uninitialised, weight_utils.py:286-300)."""
    for name in ("qweight", "qzeros"):
        p = getattr(layer, name)
        p.data = torch.randint(-2**31, 2**31 - 1, p.shape, generator=gen, device=device, dtype=torch.int64).to(
            torch.int32)
    s = layer.scales
    s.data = (torch.rand(s.shape, generator=gen, device=device) * (2e-2 - 1e-3) + 1e-3).half() * 0.1


def rms_norm(x, w, eps=1e-5):
    """RMSNorm with fp32 statistics, one fused op."""
    return F.rms_norm(x, (x.shape[-1], ), w, eps)


def rope_tables(cos, sin):
    """Full-width tables for `rope_neox` from the half-width ones: [cos, cos] and [-sin, sin]."""
    return torch.cat([cos, cos], -1), torch.cat([-sin, sin], -1)


def rope_neox(t, cos2, sin2):
    """GPT-NeoX style rotary embedding, (t1, t2) -> (t1 cos - t2 sin, t2 cos + t1 sin), as
    t * [cos, cos] + swap_halves(t) * [-sin, sin]: three elementwise launches for all heads."""
    half = t.shape[-1] // 2
    swapped = torch.cat([t[..., half:], t[..., :half]], -1)
    return torch.addcmul(t * cos2, swapped, sin2)


class DecoderLayer(torch.nn.Module):

    def __init__(self, cfg, method, max_bs, max_len, device):
        super().__init__()
        h, heads, kvh = cfg["hidden"], cfg["heads"], cfg["kv_heads"]
        tp = ps.get_tensor_model_parallel_world_size()
        self.head_dim = h // heads
        self.n_heads = heads // tp
        self.n_kv = max(1, kvh // tp)
        self.qkv_proj = QKVParallelLinear(h, self.head_dim, heads, kvh, bias=False, params_dtype=torch.float16,
                                          linear_method=method)
        self.o_proj = RowParallelLinear(h, h, bias=False, params_dtype=torch.float16, linear_method=method)
        self.gate_up_proj = MergedColumnParallelLinear(h, [cfg["inter"]] * 2, bias=False,
                                                       params_dtype=torch.float16, linear_method=method)
        self.down_proj = RowParallelLinear(cfg["inter"], h, bias=False, params_dtype=torch.float16,
                                           linear_method=method)
        self.fuse_act = True
        self.fuse_add = True
        self.ln1 = torch.ones(h, dtype=torch.float16, device=device)
        self.ln2 = torch.ones(h, dtype=torch.float16, device=device)
        self.k_cache = torch.zeros(max_bs, self.n_kv, max_len, self.head_dim, dtype=torch.float16, device=device)
        self.v_cache = torch.zeros_like(self.k_cache)

    @staticmethod
    def rms(x, w):
        return rms_norm(x, w)

    def forward(self, x, pos, cos, sin, ctx_len):
        bs = x.shape[0]
        hdn = self.rms(x, self.ln1)
        qkv, _ = self.qkv_proj(hdn)
        nq, nkv, hd = self.n_heads, self.n_kv, self.head_dim
        # rotary embedding on the query and key heads in one pass (they are adjacent in the qkv output)
        qk = rope_neox(qkv[:, :(nq + nkv) * hd].view(bs, nq + nkv, 1, hd), cos, sin)
        q, k = qk[:, :nq], qk[:, nq:]
        v = qkv[:, (nq + nkv) * hd:].view(bs, nkv, 1, hd)
        self.k_cache[:bs, :, pos:pos + 1] = k
        self.v_cache[:bs, :, pos:pos + 1] = v
        kk, vv = self.k_cache[:bs, :, :ctx_len], self.v_cache[:bs, :, :ctx_len]
        att = F.scaled_dot_product_attention(q, kk, vv, enable_gqa=self.n_kv != self.n_heads)
        if self.fuse_add:
            x = row_parallel_add_(self.o_proj, att.reshape(bs, -1), x)  # x is this step's own buffer
        else:
            o, _ = self.o_proj(att.reshape(bs, -1))
            x = x + o
        hdn = self.rms(x, self.ln2)
        if self.fuse_act:
            act = gate_up_silu_and_mul(self.gate_up_proj, hdn)  # activation in the GEMM epilogue
        else:
            gu, _ = self.gate_up_proj(hdn)
            g, u = gu.chunk(2, -1)
            act = F.silu(g) * u
        if self.fuse_add:
            return row_parallel_add_(self.down_proj, act, x)  # residual add in the GEMM epilogue
        d, _ = self.down_proj(act)
        return x + d

    def linears_only(self, x):
        """The four W4A16 GEMMs of the layer with no other work (the hot path's share of a step)."""
        a, _ = self.qkv_proj(x)
        b, _ = self.o_proj(a[:, :self.o_proj.input_size_per_partition].contiguous())
        if self.fuse_act:
            c = gate_up_silu_and_mul(self.gate_up_proj, b)
        else:
            c, _ = self.gate_up_proj(b)
            c = c[:, :self.down_proj.input_size_per_partition].contiguous()
        d, _ = self.down_proj(c)
        return d


class Model(torch.nn.Module):

    def __init__(self, cfg, n_layers, max_bs, max_len, device, seed=0, keep_stock=True):
        super().__init__()
        from vllm_quick_amd.layers.quantization.awq import AWQLinearMethod
        method = AWQLinearMethod(AWQConfig(4, 128, True), keep_stock=keep_stock)
        self.method = method
        gen = torch.Generator(device=device)
        gen.manual_seed(seed)
        self.layers = torch.nn.ModuleList()
        for _ in range(n_layers):
            layer = DecoderLayer(cfg, method, max_bs, max_len, device).to(device)
            for lin in (layer.qkv_proj, layer.o_proj, layer.gate_up_proj, layer.down_proj):
                fill_awq(lin, gen, device)
            self.layers.append(layer)
        h = cfg["hidden"]
        self.embed = (torch.randn(cfg["vocab"], h, generator=gen, device=device) * 0.05).half()
        self.lm_head = (torch.randn(h, cfg["vocab"], generator=gen, device=device) * 0.02).half()
        self.norm = torch.ones(h, dtype=torch.float16, device=device)
        hd = h // cfg["heads"]
        inv = 1.0 / (10000**(torch.arange(0, hd, 2, device=device).float() / hd))
        t = torch.arange(max_len, device=device).float()
        fr = torch.outer(t, inv)
        self.cos, self.sin = fr.cos().half(), fr.sin().half()

    def step(self, tokens, pos, ctx_len):
        x = self.embed[tokens]
        cos, sin = rope_tables(self.cos[pos], self.sin[pos])
        for layer in self.layers:
            x = layer(x, pos, cos, sin, ctx_len)
        x = DecoderLayer.rms(x, self.norm)
        return torch.argmax(x @ self.lm_head, -1)

    def linears_only(self, x):
        for layer in self.layers:
            x = layer.linears_only(x)
        return x

    def linears(self):
        for layer in self.layers:
            yield from (layer.qkv_proj, layer.o_proj, layer.gate_up_proj, layer.down_proj)

    def drop_stock(self):
        """Releases every layer's stock AWQ tensors (after one call has re-ordered them); bytes released."""
        return sum(self.method.drop_stock(lin.linear_weights) for lin in self.linears())

    def quantised_weight_bytes(self):
        """Resident bytes of the quantised linear layers: stock tensors plus private-layout copies."""
        total = 0
        for lin in self.linears():
            total += sum(t.numel() * t.element_size() for n, t in lin.linear_weights.items() if isinstance(t, torch.Tensor))
            state = lin.linear_weights.get("w4a16_packed")
            if state is not None:
                total += state[1].packed_w.numel() + state[1].packed_sz.numel()
        return total


def time_graph(fn, stream, reps=20):
    with torch.cuda.stream(stream):
        for _ in range(2):
            fn()  # warm-up: repacks every layer's weights
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        # thread-local capture mode: a process group's helper threads may call the runtime meanwhile
        mode = "thread_local" if dist.is_initialized() else "global"
        with torch.cuda.graph(g, stream=stream, capture_error_mode=mode):
            fn()
        g.replay()
        stream.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            g.replay()
            e1.record(stream)
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def run(model_name="7b", batch_sizes=(1, 2, 4, 8, 16, 32, 64), n_layers=None, ctx_len=128, device=None,
        fuse_act=True, fuse_add=True, keep_stock=True, rehearse_tp=0):
    """`rehearse_tp` = N > 1 on a single process: the step of ONE rank of an N-way tensor-parallel model (layers
    sharded as rank 0's, every all-reduce a device copy of the same size): the compute half of a tensor-parallel
    step, to which a run on N GPUs adds the collectives."""
    cfg = SHAPES[model_name]
    device = device or torch.device("cuda", torch.cuda.current_device())
    n_layers = n_layers or cfg["layers"]
    rehearsal = ps.rehearse_one_rank(rehearse_tp) if rehearse_tp > 1 else contextlib.nullcontext()
    with torch.inference_mode(), rehearsal:
        model = Model(cfg, n_layers, max(batch_sizes), ctx_len + 8, device, keep_stock=keep_stock)
        for layer in model.layers:
            layer.fuse_act = fuse_act
            layer.fuse_add = fuse_add
        stream = torch.cuda.Stream(device=device)
        res = {}
        model.linears_only((torch.randn(1, cfg["hidden"], device=device) * 0.25).half())  # re-orders every layer
        bytes_both = model.quantised_weight_bytes()
        released = model.drop_stock()
        for bs in batch_sizes:
            tokens = torch.randint(0, cfg["vocab"], (bs, ), device=device)
            ms = time_graph(lambda: model.step(tokens, ctx_len, ctx_len + 1), stream)
            xin = (torch.randn(bs, cfg["hidden"], device=device) * 0.25).half()
            ms_lin = time_graph(lambda: model.linears_only(xin), stream)
            scale = cfg["layers"] / n_layers  # extrapolate a reduced-depth run to the full model
            res[str(bs)] = {
                "ms_per_step": ms * scale,
                "tokens_per_s": bs / (ms * scale * 1e-3),
                "w4a16_gemm_ms": ms_lin * scale,
                "w4a16_share": ms_lin / ms,
            }
    return {"model": f"llama-2-{model_name}-awq-shaped (synthetic weights, g128)", "layers_run": n_layers,
            "layers_model": cfg["layers"], "context": ctx_len,
            "tp": rehearse_tp if rehearse_tp > 1 else ps.get_tensor_model_parallel_world_size(),
            "ranks_run": 1 if rehearse_tp > 1 else ps.get_tensor_model_parallel_world_size(),
            "all_reduce": "a device copy of the same size (one rank run alone)" if rehearse_tp > 1 else None,
            "gate_up_activation": "fused into the GEMM epilogue" if fuse_act else "separate elementwise ops",
            "residual_add": "fused into the o_proj / down_proj GEMM epilogue (one rank)" if fuse_add else "separate op",
            "quantised_weight_bytes_resident": int((bytes_both - released) * cfg["layers"] / n_layers),
            "quantised_weight_bytes_with_stock_tensors": int(bytes_both * cfg["layers"] / n_layers),
            "stock_tensors": "kept beside the private layout" if keep_stock else "released after re-ordering",
            "by_batch": res}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="7b", choices=sorted(SHAPES))
    ap.add_argument("--bs", default="1,2,4,8,16,32,64")
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--ctx", type=int, default=128)
    ap.add_argument("--no-fuse-act", action="store_true", help="run gate_up GEMM and silu-and-mul as separate ops")
    ap.add_argument("--no-fuse-add", action="store_true", help="add the o_proj / down_proj outputs to the residual "
                    "stream as separate ops")
    ap.add_argument("--drop-stock", action="store_true", help="release the stock AWQ tensors once re-ordered")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl (= RCCL), or gloo for a "
                    "rehearsal (then every all-reduce must take the peer-to-peer path)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--p2p-all-reduce", action="store_true",
                    help="tensor parallel: the one-kernel peer-to-peer all-reduce instead of RCCL")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if a.same_device else int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(a.backend)
        ps.initialize_model_parallel(world)
        if a.p2p_all_reduce:
            from vllm_quick_amd.parallel_utils import custom_all_reduce as car
            car.init_custom_ar()
        elif a.backend == "nccl":
            # a decode step is captured in a HIP graph, where torch.distributed's RCCL path cannot
            # be used: a communicator of this harness's own (vllm/worker/worker.py:257-262 sets up
            # the reference's CuPy one the same way)
            import socket
            from vllm_quick_amd.parallel_utils import rccl_utils
            port = [0]
            if dist.get_rank() == 0:
                with socket.socket() as sock:
                    sock.bind(("127.0.0.1", 0))
                    port[0] = sock.getsockname()[1]
            dist.broadcast_object_list(port, src=0)
            rccl_utils.init_process_group(world, dist.get_rank(), "127.0.0.1", port[0])
    t0 = time.time()
    from vllm_quick_amd.parallel_utils import rccl_utils as direct
    with (ps.with_rccl_for_all_reduce() if direct.is_initialized() else contextlib.nullcontext()):
        out = run(a.model, tuple(int(v) for v in a.bs.split(",")), a.layers or None, a.ctx,
                  fuse_act=not a.no_fuse_act, fuse_add=not a.no_fuse_add, keep_stock=not a.drop_stock)
    out["wall_s"] = time.time() - t0
    out["all_reduce"] = "peer-to-peer kernel" if world > 1 and a.p2p_all_reduce else ("rccl" if world > 1 else None)
    if world > 1 and a.p2p_all_reduce:
        from vllm_quick_amd.parallel_utils import custom_all_reduce as car
        out["all_reduce_timed_out"] = car.get_handle().timed_out() if car.get_handle() is not None else None
        car.destroy_custom_ar()
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps(out))
    if world > 1:
        direct.destroy_process_group()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
