"""Mixtral-8x7B-shaped quantised expert block (8 experts x {w1, w3: 4096 -> 14336, w2: 14336 -> 4096},
AWQ g128, synthetic seeded weights): the reference's per-expert loop of single W4A16 launches
(mixtral_quant.py:145-157) against three batched launches, both captured in a HIP graph.

    python benchmarks/mixtral_experts.py [--tokens 1,2,4,8,16] [--experts 8]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from benchmarks.llama_decode import fill_awq, time_graph  # noqa: E402
from vllm_quick_amd.layers.experts import MixtralMoE  # noqa: E402
from vllm_quick_amd.layers.quantization import AWQConfig  # noqa: E402


def run(tokens=(1, 2, 4, 8, 16), experts=8, hidden=4096, inter=14336, top_k=2, device=None):
    device = device or torch.device("cuda", torch.cuda.current_device())
    method = AWQConfig(4, 128, True).get_linear_method()
    gen = torch.Generator(device=device)
    gen.manual_seed(0)
    res = {}
    with torch.inference_mode():
        moe = MixtralMoE(experts, top_k, hidden, inter, linear_method=method, params_dtype=torch.float16).to(device)
        for idx in moe.expert_indicies:
            for name in ("w1", "w2", "w3"):
                fill_awq(getattr(moe.experts[idx], name), gen, device)
        moe.gate.weight.data = (torch.randn(moe.gate.weight.shape, generator=gen, device=device) * 0.05).half()
        stream = torch.cuda.Stream(device=device)
        wbytes = experts * 3 * (hidden * inter // 2 + (hidden // 128) * inter * 2 + (hidden // 128) * inter // 2)
        for t in tokens:
            x = (torch.randn(1, t, hidden, generator=gen, device=device) * 0.25).half()
            moe.batched = True
            ms_b = time_graph(lambda: moe(x), stream)
            keep, moe.sparse_max_tokens = moe.sparse_max_tokens, 0
            ms_all = time_graph(lambda: moe(x), stream)  # every expert runs, as in the reference
            moe.sparse_max_tokens = keep
            moe.batched = False
            ms_l = time_graph(lambda: moe(x), stream)
            active = int((moe.routing(x.view(-1, hidden)) != 0).any(0).sum())
            res[str(t)] = {"batched_ms": ms_b, "batched_all_experts_ms": ms_all, "loop_ms": ms_l,
                           "speedup": ms_l / ms_b, "experts_selected": active,
                           "experts_run": active if t <= keep else experts,
                           "batched_weight_GBps": wbytes * (active if t <= keep else experts) / experts /
                           (ms_b * 1e-3) * 1e-9}
    return {"block": f"mixtral-8x7b-shaped experts (E={experts}, {hidden}->{inter}->{hidden}, g128, synthetic)",
            "weight_bytes": wbytes, "by_tokens": res}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", default="1,2,4,8,16")
    ap.add_argument("--experts", type=int, default=8)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    print(json.dumps(run(tuple(int(v) for v in a.tokens.split(",")), a.experts)))
