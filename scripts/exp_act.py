"""Launch-shape sweep of the fused gate_up + silu-and-mul launch (developer script)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
K, N, G = 4096, 22016, 128
Ms = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,8,16,32").split(",")]
packs = [ops.awq_repack(*make_weights(K, N, G, i, dev)) for i in range(9)]
for M in Ms:
    x = make_x(M, K, 1, dev)
    mt = -(-M // 16)
    d = time_graph([(lambda p=p: ops.awq_gemm_packed_silu_and_mul(x, p)) for p in packs])
    res = []
    for cfg in [(mt, 2, 16), (mt, 2, 8), (mt, 4, 8), (mt, 6, 8), (mt, 8, 8), (mt, 2, 4), (mt, 4, 4)]:
        try:
            us = time_graph([(lambda p=p: ops.awq_gemm_packed_silu_and_mul(x, p, cfg=cfg)) for p in packs])
            res.append((us, cfg))
        except Exception:
            pass
    res.sort()
    print(f"gate_up+act M={M}: default {d:.2f} us | " + "  ".join(f"{c}:{us:.2f}" for us, c in res), flush=True)
