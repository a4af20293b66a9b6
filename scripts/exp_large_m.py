"""Fused kernel at many rows (developer script): run once per library build."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
tag = os.path.basename(os.environ.get("W4A16_LIBRARY", "default"))
G = 128
for K, N in ((4096, 11008), (4096, 22016), (11008, 4096)):
    packs = [ops.awq_repack(*make_weights(K, N, G, i, dev)) for i in range(8)]
    for M in (640, 1024, 2048, 4096):
        x = make_x(M, K, 1, dev)
        res = []
        for cfg in (None, (4, 3, 4, 2), (4, 3, 2, 4)):
            us = time_graph([(lambda p=p: ops.awq_gemm_packed(x, p, cfg=cfg)) for p in packs], iters=10)
            res.append(f"{cfg}:{us:.1f} us ({2.0*M*K*N/us*1e-6:.0f} TF)")
        print(f"{tag} {K}x{N} M={M}: " + "  ".join(res), flush=True)
    del packs
    torch.cuda.empty_cache()
