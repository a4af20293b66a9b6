"""Every built launch shape that can serve up to 32 rows, against the default choice (developer script)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
K, N = int(sys.argv[1]), int(sys.argv[2])
nb = max(4, min(16, int(400e6 / (K * N / 2))))
packs = [ops.awq_repack(*make_weights(K, N, 128, i, dev)) for i in range(nb)]
stream = {1: [(1, t, w) for t, w in ((1, 16), (2, 16), (3, 16), (1, 8), (2, 8), (3, 8), (4, 8), (6, 8), (1, 4), (2, 4), (3, 4), (4, 4))],
          2: [(2, t, w) for t, w in ((1, 16), (1, 8), (2, 8), (3, 8), (4, 8), (1, 4), (2, 4), (3, 4))]}
tiled = {1: [(1, 3, 1, 8), (1, 1, 3, 4), (1, 3, 1, 4)], 2: [(2, 3, 1, 8), (2, 3, 1, 4), (2, 3, 2, 4), (2, 6, 1, 4)]}
for M in (1, 4, 8, 16, 24, 32):
    x = make_x(M, K, 1, dev)
    mt = 1 if M <= 16 else 2
    res = []
    for cfg in [None] + stream[mt] + tiled[mt]:
        try:
            us = min(time_graph([(lambda p=p: ops.awq_gemm_packed(x, p, cfg=cfg)) for p in packs], iters=20) for _ in range(2))
            res.append((us, cfg))
        except Exception:
            pass
    base = [u for u, c in res if c is None][0]
    res.sort()
    print("JSON " + json.dumps({"K": K, "N": N, "M": M, "us": {str(c): u for u, c in res}}), flush=True)
    print(f"{K}x{N} M={M}: default {base:.2f} us | " + "  ".join(f"{c}:{u:.2f}" for u, c in res[:5]), flush=True)
