"""Standalone dequantise kernels: time and bytes moved (developer script)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights
dev = torch.device("cuda:0")
G = 128
for K, N in ((4096, 11008), (11008, 4096), (4096, 22016)):
    ws = [make_weights(K, N, G, i, dev) for i in range(6)]
    packs = [ops.awq_repack(*w) for w in ws]
    by = K * N // 2 + (K // G) * N * 2 + (K // G) * N // 2 + 2 * K * N
    a = time_graph([(lambda w=w: ops.awq_dequantize(w[0], w[1], w[2], 0, 0, 0)) for w in ws])
    b = time_graph([(lambda p=p: ops.awq_dequantize_packed(p)) for p in packs])
    print(f"{K}x{N}: stock {a:.1f} us ({by/a*1e-6:.2f} TB/s)  packed {b:.1f} us ({by/b*1e-6:.2f} TB/s)", flush=True)
