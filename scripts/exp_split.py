"""K split of the decode kernel on narrow layers: (t, 16 waves) x splits, graph-replayed (developer script)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
ws = ops.gemm_workspace(dev, 1 << 20)
for K, N in ((8192, 1280), (8192, 512), (14336, 512), (11008, 1024), (4096, 512), (16384, 1024), (8192, 2048), (4096, 1376), (4096, 4096), (11008, 4096), (8192, 3584)):
    nb = 16
    packs = [ops.awq_repack(*make_weights(K, N, 128, i, dev)) for i in range(nb)]
    for M in (1, 4):
        x = make_x(M, K, 1, dev)
        res = []
        for d in (1, 2):
            for s in (1, 2, 3, 4):
                t = 1
                cfg = ("decode", t, 16, s, d)
                try:
                    us = min(time_graph([(lambda p=p: ops.awq_gemm_packed(x, p, cfg=cfg, workspace=ws)) for p in packs], iters=20) for _ in range(3))
                except Exception:
                    continue
                res.append(f"d{d}s{s}:{us:.2f}")
        print(f"{K}x{N} M={M}: " + " ".join(res), flush=True)
