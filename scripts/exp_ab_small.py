"""A/B timing of the default launch at decode row counts on the Llama-2-7B layer shapes (developer script):
run once per library (W4A16_LIBRARY)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
tag = os.path.basename(os.environ.get("W4A16_LIBRARY", "default"))
shapes = [(4096, 11008), (4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for K, N in shapes:
    nb = max(4, min(16, int(400e6 / (K * N / 2))))
    packs = [ops.awq_repack(*make_weights(K, N, 128, i, dev)) for i in range(nb)]
    row = []
    for M in (1, 4, 8, 16, 32):
        x = make_x(M, K, 1, dev)
        us = min(time_graph([(lambda p=p: ops.awq_gemm_packed(x, p)) for p in packs], iters=40) for _ in range(2))
        row.append(f"M={M}: {us:6.2f}")
    print(f"{tag:24s} {K}x{N}  " + "  ".join(row), flush=True)
    del packs
