"""1-row-tile shapes of the LDS-staged kernel against the streaming kernel at 9-32 rows (developer script)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
G = 128
for K, N in ((4096, 11008), (4096, 4096), (11008, 4096)):
    nsets = max(2, int(400e6 // (K * N // 2)) + 1)
    packs = [ops.awq_repack(*make_weights(K, N, G, i, dev)) for i in range(nsets)]
    for M, cfgs in ((16, [(1, 3, 1, 8), (1, 1, 3, 4), (1, 3, 1, 4)]), (32, [(2, 3, 1, 8), (2, 3, 1, 4), (2, 3, 2, 4)])):
        x = make_x(M, K, 1, dev)
        d = time_graph([(lambda p=p: ops.awq_gemm_packed(x, p)) for p in packs])
        out = []
        for cfg in cfgs:
            us = time_graph([(lambda p=p: ops.awq_gemm_packed(x, p, cfg=cfg)) for p in packs])
            ref = ops.awq_gemm_packed(x, packs[0])
            got = ops.awq_gemm_packed(x, packs[0], cfg=cfg)
            out.append(f"{cfg}:{us:.2f} (maxdiff {(ref.float()-got.float()).abs().max().item():.1e})")
        print(f"{K}x{N} M={M}: default {d:.2f} | " + "  ".join(out), flush=True)
    del packs
    torch.cuda.empty_cache()
