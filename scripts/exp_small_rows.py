"""Developer aid: default launch per layer shape at 17-32 rows (run once per library build and compare)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_weights, make_x
from scripts._timing import time_graph
from vllm_quick_amd import _lib
from vllm_quick_amd._C import ops

dev = torch.device("cuda:0")
G = 128
shapes = [(4096, 11008), (4096, 12288), (4096, 4096), (11008, 4096), (4096, 22016), (8192, 1280), (1024, 8192),
          (8192, 7168), (3584, 8192), (4096, 1376), (4096, 2752), (4096, 5504), (2048, 11008), (1024, 11008),
          (4096, 14336), (14336, 4096)]
for K, N in shapes:
    n = max(4, min(16, (400 << 20) // (K * N // 2)))
    packs = [ops.awq_repack(*make_weights(K, N, G, i, dev)) for i in range(n)]
    out = []
    for M in (17, 24, 32):
        x = make_x(M, K, 1, dev)
        us = min(time_graph([(lambda p=p: ops.awq_gemm_packed(x, p)) for p in packs], iters=20) for _ in range(3))
        out.append(f"M={M} {us:6.2f} {_lib.launch_shape(M, K, N)}")
    print(f"{K}x{N}: " + "  ".join(out), flush=True)
    del packs
