"""The default launch of two builds of the library side by side, in separate processes (developer script).
Usage: exp_defaults.py LIB KxN... ; prints one line per shape and row count."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from vllm_quick_amd import _lib
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for K, N in shapes:
    nb = max(4, min(16, int(400e6 / (K * N / 2))))
    packs = [ops.awq_repack(*make_weights(K, N, 128, i, dev)) for i in range(nb)]
    out = []
    for M in (1, 4, 8, 16):
        x = make_x(M, K, 1, dev)
        fns = [(lambda p=p: ops.awq_gemm_packed(x, p)) for p in packs]
        time_graph(fns, iters=20)
        us = min(time_graph(fns, iters=20) for _ in range(3))
        out.append(f"M={M}: {us:.2f} {_lib.launch_shape(M, K, N)}")
    print(f"{K}x{N} " + " | ".join(out), flush=True)
