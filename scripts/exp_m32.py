import os, sys, torch
sys.path.insert(0, "/root/repo")
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
for K, N in ((4096, 4096), (8192, 1280), (4096, 8192), (2048, 4096)):
    packs = [ops.awq_repack(*make_weights(K, N, 128, i, dev)) for i in range(16)]
    for M in (26, 28, 30, 32):
        x = make_x(M, K, 1, dev)
        r = []
        for cfg in (None, (2, 1, 16), (2, 1, 8), (2, 2, 8), (2, 3, 1, 8), (2, 3, 1, 4)):
            try:
                r.append((min(time_graph([(lambda p=p: ops.awq_gemm_packed(x, p, cfg=cfg)) for p in packs], iters=20) for _ in range(3)), cfg))
            except Exception:
                pass
        print(f"{K}x{N} M={M}: " + "  ".join(f"{c}:{u:.2f}" for u, c in r), flush=True)
