"""A/B of the default launch on the Llama-2-7B layer shapes at a few row counts (developer script): run once
per library (W4A16_LIBRARY=...)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
shapes = {"qkv": (4096, 12288), "o": (4096, 4096), "gate_up": (4096, 22016), "down": (11008, 4096)}
Ms = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,8").split(",")]
tag = os.path.basename(os.environ.get("W4A16_LIBRARY", "default"))
for name, (K, N) in shapes.items():
    nsets = max(2, int(400e6 // (K * N // 2)) + 1)
    packs = [ops.awq_repack(*make_weights(K, N, 128, i, dev)) for i in range(nsets)]
    for M in Ms:
        x = make_x(M, K, 1, dev)
        ts = [time_graph([(lambda p=p: ops.awq_gemm_packed(x, p)) for p in packs]) for _ in range(3)]
        print(f"{tag} {name} M={M}: " + " ".join(f"{t:.2f}" for t in ts), flush=True)
    del packs
    torch.cuda.empty_cache()
