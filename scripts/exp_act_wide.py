"""Developer aid: merged gate_up GEMM with the activation fused (16x16x32 kernels) against the plain GEMM on the
default launch (32x32x16 kernels from 21 rows) plus a separate activation step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from bench import make_weights, make_x
from scripts._timing import time_graph
from vllm_quick_amd._C import ops

dev = torch.device("cuda:0")
G = 128
for K, N in ((4096, 22016), (8192, 7168), (4096, 28672)):
    packs = [ops.awq_repack(*make_weights(K, N, G, i, dev)) for i in range(8)]
    for M in (16, 24, 32, 64, 128, 256, 512):
        x = make_x(M, K, 1, dev)

        def two_step(p):
            h = ops.awq_gemm_packed(x, p)
            return F.silu(h[:, :N // 2]) * h[:, N // 2:]
        fused = min(time_graph([(lambda p=p: ops.awq_gemm_packed_silu_and_mul(x, p)) for p in packs], iters=20) for _ in range(2))
        plain = min(time_graph([(lambda p=p: ops.awq_gemm_packed(x, p)) for p in packs], iters=20) for _ in range(2))
        both = min(time_graph([(lambda p=p: two_step(p)) for p in packs], iters=20) for _ in range(2))
        print(f"{K}x{N} M={M:4d}: fused {fused:7.2f}  plain gemm {plain:7.2f}  gemm + activation {both:7.2f}", flush=True)
    del packs
