"""Default launch of one layer shape at several row counts (run once per library build, W4A16_LIBRARY, on the same
box for an A/B).  Usage: exp_ab_default.py K N M[,M..]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
K, N = int(sys.argv[1]), int(sys.argv[2])
Ms = [int(v) for v in sys.argv[3].split(",")]
nb = max(4, min(16, int(400e6 / (K * N / 2))))
packs = [ops.awq_repack(*make_weights(K, N, 128, i, dev)) for i in range(nb)]
tag = os.path.basename(os.environ.get("W4A16_LIBRARY", "default"))
for M in Ms:
    x = make_x(M, K, 1, dev)
    us = min(time_graph([(lambda p=p: ops.awq_gemm_packed(x, p)) for p in packs], iters=30) for _ in range(3))
    print(f"{tag:24s} {K}x{N} M={M}: {us:.2f} us", flush=True)
