"""Fused gate_up + silu_and_mul launch: every built streaming shape against the default (developer script)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
K, N = int(sys.argv[1]), int(sys.argv[2])
nb = max(4, min(16, int(400e6 / (K * N / 2))))
packs = [ops.awq_repack(*make_weights(K, N, 128, i, dev)) for i in range(nb)]
cfgs = {1: [None, (1, 2, 16), (1, 2, 8), (1, 4, 8), (1, 6, 8), (1, 8, 8), (1, 2, 4), (1, 4, 4)],
        2: [None, (2, 2, 8), (2, 4, 8), (2, 6, 8), (2, 2, 4), (2, 3, 2, 4)]}
for M in (1, 4, 8, 16, 32):
    x = make_x(M, K, 1, dev)
    res = []
    for cfg in cfgs[1 if M <= 16 else 2]:
        try:
            us = min(time_graph([(lambda p=p: ops.awq_gemm_packed_silu_and_mul(x, p, cfg=cfg)) for p in packs], iters=20)
                     for _ in range(3))
            res.append((us, cfg))
        except Exception as e:
            pass
    base = [u for u, c in res if c is None][0]
    res.sort()
    print(f"act {K}x{N} M={M}: default {base:.2f} us | " + "  ".join(f"{c}:{u:.2f}" for u, c in res[:6]), flush=True)
