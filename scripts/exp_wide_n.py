"""Decode kernel shapes on very wide layers at 1-4 rows against the default launch (developer script)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(4096, 22016), (4096, 28672), (4096, 14336), (8192, 28672), (5120, 27648)]
for K, N in shapes:
    nb = max(4, min(16, int(400e6 / (K * N / 2))))
    packs = [ops.awq_repack(*make_weights(K, N, 128, i, dev)) for i in range(nb)]
    for M in (1, 4):
        x = make_x(M, K, 1, dev)
        ref = ops.awq_gemm_packed(x, packs[0]).float()
        res = []
        for cfg in [None] + [("decode", t, w, 1, d) for t, w in ((1, 16), (2, 16), (3, 16), (6, 16), (1, 8), (2, 8), (3, 8), (4, 8), (6, 8), (8, 8)) for d in (1, 2)]:
            try:
                y = ops.awq_gemm_packed(x, packs[0], cfg=cfg).float()
            except Exception:
                continue
            err = (y - ref).abs().max().item()
            us = min(time_graph([(lambda p=p: ops.awq_gemm_packed(x, p, cfg=cfg)) for p in packs], iters=20) for _ in range(3))
            res.append((us, cfg, err))
        base = res[0][0]
        res.sort(key=lambda r: r[0])
        print(f"{K}x{N} M={M}: default {base:.2f} | " + "  ".join(f"{c[1:] if c else c}:{u:.2f}({e:.0e})" for u, c, e in res[:8]), flush=True)
