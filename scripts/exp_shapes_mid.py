"""Launch-shape sweep at 17-128 rows on Llama-2-7B layer shapes (developer script)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
shapes = {"qkv": (4096, 12288), "o": (4096, 4096), "gate_up": (4096, 22016), "down": (11008, 4096)}
if len(sys.argv) > 2 and sys.argv[2] == "tp":  # the shard shapes of a 4096 x 11008 layer over 2/4/8 ranks
    shapes = {f"col{p}": (4096, 11008 // p) for p in (2, 4, 8)}
    shapes.update({f"row{p}": (4096 // p, 11008) for p in (2, 4, 8)})
Ms = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "32,64").split(",")]
G = 128
if sys.argv[-1] == "70b":  # Llama-2-70B layers as one of 8 tensor-parallel ranks holds them
    shapes = {"qkv": (8192, 1280), "o": (1024, 8192), "gate_up": (8192, 7168), "down": (3584, 8192)}
STREAM = [(2, 1, 16), (2, 1, 8), (2, 2, 8), (2, 3, 8), (2, 4, 8), (2, 6, 8), (2, 1, 4), (2, 2, 4), (2, 3, 4),
          (4, 1, 8), (4, 2, 8), (4, 3, 8), (4, 4, 8), (4, 6, 8), (4, 2, 4), (4, 3, 4)]
TILED = [(4, 3, 2, 4), (4, 3, 4, 2), (4, 3, 1, 4), (4, 2, 2, 4), (4, 2, 3, 4), (2, 3, 2, 4), (2, 3, 1, 4), (4, 3, 2, 2),
         (2, 6, 1, 4), (4, 6, 1, 4), (2, 3, 1, 8)]
for name, (K, N) in shapes.items():
    nsets = min(64, max(2, int(400e6 // (K * N // 2)) + 1))
    packs = [ops.awq_repack(*make_weights(K, N, G, i, dev)) for i in range(nsets)]
    for M in Ms:
        x = make_x(M, K, 1, dev)
        mt = -(-M // 16)
        res = []
        d = time_graph([(lambda p=p: ops.awq_gemm_packed(x, p)) for p in packs])
        for cfg in [c for c in STREAM if c[0] >= mt or M > 64] + TILED:
            if len(cfg) == 3 and cfg[0] * 16 < M and M <= 64:
                continue
            try:
                us = time_graph([(lambda p=p: ops.awq_gemm_packed(x, p, cfg=cfg)) for p in packs])
                res.append((us, cfg))
            except Exception:
                pass
        res.sort()
        fl = 2.0 * M * K * N
        print(f"{name} {K}x{N} M={M}: default {d:.2f} us ({fl/d*1e-6:.0f} TF) | best " +
              "  ".join(f"{c}:{us:.2f}" for us, c in res[:5]), flush=True)
    del packs
    torch.cuda.empty_cache()
