"""The 32x32x16 kernel's launch shapes against the default choice, with a bit-level check against the
default on the same inputs (developer script).  usage: exp_wide.py [K N] [rows ...]"""
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
rows = args or [64, 128, 256, 512, 1024]
packs = [ops.awq_repack(*make_weights(K, N, G, i, dev)) for i in range(16)]
wide = [("wide", 2, 3, 8), ("wide", 2, 2, 8), ("wide", 2, 1, 8), ("wide", 1, 3, 8), ("wide", 1, 2, 8), ("wide", 2, 3, 4),
        ("wide", 2, 2, 4), ("wide", 1, 3, 4)]
for M in rows:
    x = make_x(M, K, 1, dev)
    ref = ops.awq_gemm_packed(x, packs[0]).float()
    res = []
    for cfg in [None] + wide:
        try:
            y = ops.awq_gemm_packed(x, packs[0], cfg=cfg).float()
        except Exception as e:
            print(cfg, "refused:", e)
            continue
        err = (y - ref).abs().max().item()
        us = min(time_graph([(lambda p=p: ops.awq_gemm_packed(x, p, cfg=cfg)) for p in packs], iters=20) for _ in range(2))
        res.append((us, cfg, err))
    print(f"{K}x{N} M={M}: " + "  ".join(f"{c}:{u:.1f}us(d{e:.1e})" for u, c, e in res), flush=True)
    print("JSON " + json.dumps({"K": K, "N": N, "M": M, "us": {str(c): u for u, c, e in res}}), flush=True)
