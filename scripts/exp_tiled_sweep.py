"""Every built launch shape of the LDS-staged kernel at a few row counts (developer script)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
K, N, G = 4096, 11008, 128
if len(sys.argv) > 2:
    K, N = int(sys.argv[1]), int(sys.argv[2])
packs = [ops.awq_repack(*make_weights(K, N, G, i, dev)) for i in range(16)]
cfgs = [None, (4, 2, 3, 4), (4, 3, 2, 4), (4, 3, 1, 4), (4, 2, 2, 4), (4, 3, 2, 2), (4, 3, 4, 2), (2, 3, 2, 4), (2, 3, 1, 4),
        (8, 3, 1, 2), (8, 3, 2, 2), (4, 6, 1, 4), (2, 6, 1, 4), (2, 3, 1, 8),
        # streaming kernel: (row tiles, column tiles, waves)
        (4, 1, 8), (4, 2, 8), (4, 3, 8), (4, 4, 8), (2, 2, 8), (2, 3, 8), (2, 4, 8), (8, 1, 8), (8, 2, 8), (8, 3, 8), (4, 2, 4), (4, 3, 4)]
for M in (48, 64, 128, 256, 512):
    x = make_x(M, K, 1, dev)
    res = []
    for cfg in cfgs:
        try:
            us = min(time_graph([(lambda p=p: ops.awq_gemm_packed(x, p, cfg=cfg)) for p in packs], iters=20) for _ in range(2))
            res.append((us, cfg))
        except Exception as e:  # a shape the entry point refuses for this layer
            pass
    base = [u for u, c in res if c is None][0]
    res.sort()
    import json
    print("JSON " + json.dumps({"K": K, "N": N, "M": M, "us": {str(c): u for u, c in res}}), flush=True)
    print(f"{K}x{N} M={M}: default {base:.1f} us | " + "  ".join(f"{c}:{u:.1f}" for u, c in res[:6]), flush=True)
