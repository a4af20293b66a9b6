"""Decode kernel shapes on long-K layers at 1-4 rows: 8 or 16 waves, 1 or 2 blocks in flight (developer script)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
for K, N in ((14336, 4096), (14336, 512), (16384, 1024), (28672, 1024), (12800, 4096), (11008, 4096)):
    nb = max(4, min(16, int(400e6 / (K * N / 2))))
    packs = [ops.awq_repack(*make_weights(K, N, 128, i, dev)) for i in range(nb)]
    for M in (1, 2, 4):
        x = make_x(M, K, 1, dev)
        res = []
        for cfg in (None, ("decode", 1, 16, 1, 1), ("decode", 1, 16, 1, 2), ("decode", 1, 8, 1, 1), ("decode", 1, 8, 1, 2), ("decode", 2, 8, 1, 2), (1, 1, 8), (1, 1, 16)):
            try:
                us = min(time_graph([(lambda p=p: ops.awq_gemm_packed(x, p, cfg=cfg)) for p in packs], iters=20) for _ in range(3))
            except Exception:
                continue
            res.append(f"{cfg}:{us:.2f}")
        print(f"{K}x{N} M={M}: " + "  ".join(res), flush=True)
