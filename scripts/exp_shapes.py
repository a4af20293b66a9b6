"""Launch-shape sweep of the streaming kernel on Llama-2-7B layer shapes (developer script)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
shapes = {"qkv": (4096, 12288), "o": (4096, 4096), "gate_up": (4096, 22016), "down": (11008, 4096)}
Ms = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,8").split(",")]
G = 128
if sys.argv[-1] == "70b":  # Llama-2-70B layers as one of 8 tensor-parallel ranks holds them
    shapes = {"qkv": (8192, 1280), "o": (1024, 8192), "gate_up": (8192, 7168), "down": (3584, 8192)}
CFG = [(1, 1, 16), (1, 2, 16), (1, 3, 16), (1, 1, 8), (1, 2, 8), (1, 3, 8), (1, 4, 8), (1, 6, 8), (1, 8, 8),
       (1, 1, 4), (1, 2, 4), (1, 3, 4), (1, 4, 4)]
for name, (K, N) in shapes.items():
    nsets = max(2, int(400e6 // (K * N // 2)) + 1)
    packs = [ops.awq_repack(*make_weights(K, N, G, i, dev)) for i in range(nsets)]
    by = K * N // 2 + (K // G) * N * 2 + (K // G) * N // 2
    for M in Ms:
        x = make_x(M, K, 1, dev)
        res = []
        d = time_graph([(lambda p=p: ops.awq_gemm_packed(x, p)) for p in packs])
        for cfg in CFG:
            c = (-(-M // 16),) + cfg[1:]
            try:
                us = time_graph([(lambda p=p: ops.awq_gemm_packed(x, p, cfg=c)) for p in packs])
                res.append((us, c))
            except Exception:
                pass
        res.sort()
        print(f"{name} {K}x{N} M={M}: default {d:.2f} us ({by/d*1e-6:.2f} TB/s) | best " +
              "  ".join(f"{c}:{us:.2f}" for us, c in res[:4]), flush=True)
    del packs
    torch.cuda.empty_cache()
