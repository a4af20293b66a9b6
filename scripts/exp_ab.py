"""A/B timing of the default launch at a few row counts (developer script): run once per library."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
K, N, G = 4096, 11008, 128
packs = [ops.awq_repack(*make_weights(K, N, G, i, dev)) for i in range(16)]
tag = os.path.basename(os.environ.get("W4A16_LIBRARY", "default"))
for rep in range(2):
    for M in (1, 16, 32, 128, 512):
        x = make_x(M, K, 1, dev)
        us = time_graph([(lambda p=p: ops.awq_gemm_packed(x, p)) for p in packs], iters=40)
        print(f"{tag} rep{rep} M={M}: {us:.2f} us", flush=True)
