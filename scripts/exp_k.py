import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
for K in (4096, 4224, 3968, 8192, 8320):
    N, G = 11008, 128
    packs = []
    for i in range(16):
        qw, s, qz = make_weights(K, N, G, i, dev)
        packs.append(ops.awq_repack(qw, s, qz))
    for M, cfg in ((128, (8, 3, 8)), (128, (4, 6, 8)), (16, (1, 3, 16)), (1, (1, 3, 16))):
        x = make_x(M, K, 1, dev)
        us = time_graph([(lambda p=p: ops.awq_gemm_packed(x, p, cfg=cfg)) for p in packs])
        print(f"K={K} M={M} cfg={cfg}: {us:.2f} us  per-kblock {us / (K / 128) * 1e3:.0f} ns", flush=True)
