"""The shared-slab 32x32x16 kernel's launch shapes at many rows against the default choice and against
dequantise + torch.matmul (developer script).  usage: exp_shared.py [K N] [rows ...]"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
K, N, G = 4096, 11008, 128
args = [int(a) for a in sys.argv[1:]]
if len(args) >= 2:
    K, N = args[:2]
    args = args[2:]
rows = args or [256, 512, 1024, 2048, 4096]
packs = [ops.awq_repack(*make_weights(K, N, G, i, dev)) for i in range(8)]
shared = [("wide", 4, 1, -8), ("wide", 4, 2, -8), ("wide", 2, 3, -8), ("wide", 4, 3, -4), ("wide", 4, 2, -4), ("wide", 2, 2, -8)]
for M in rows:
    x = make_x(M, K, 1, dev)
    ref = ops.awq_gemm_packed(x, packs[0], cfg=(4, 3, 4, 2)).float()
    res = []
    for cfg in [None] + shared:
        try:
            y = ops.awq_gemm_packed(x, packs[0], cfg=cfg).float()
        except Exception as e:
            print(cfg, "refused:", e)
            continue
        err = (y - ref).abs().max().item()
        us = min(time_graph([(lambda p=p: ops.awq_gemm_packed(x, p, cfg=cfg)) for p in packs], iters=10) for _ in range(2))
        res.append((us, cfg, err))
    def dq(p):
        return x @ ops.awq_dequantize_packed(p)
    us = min(time_graph([(lambda p=p: dq(p)) for p in packs], iters=10) for _ in range(2))  # per call: two launches
    res.append((us, "dequant+matmul", 0.0))
    tf = 2.0 * M * K * N * 1e-6
    print(f"{K}x{N} M={M}: " + "  ".join(f"{c}:{u:.1f}us={tf / u:.0f}TF(d{e:.0e})" for u, c, e in res), flush=True)
    print("JSON " + json.dumps({"K": K, "N": N, "M": M, "us": {str(c): u for u, c, e in res}}), flush=True)
