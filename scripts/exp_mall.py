"""Same weight set on every launch (resident in the 256 MB memory-side cache after the first) against 16
rotating sets (always from HBM): what a prefetch of the next layer's weights could buy at best (developer script)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
for K, N in ((4096, 11008), (4096, 22016), (11008, 4096), (4096, 4096)):
    packs = [ops.awq_repack(*make_weights(K, N, 128, i, dev)) for i in range(16)]
    for M in (1, 8, 128):
        x = make_x(M, K, 1, dev)
        rot = min(time_graph([(lambda p=p: ops.awq_gemm_packed(x, p)) for p in packs], iters=40) for _ in range(2))
        same = min(time_graph([(lambda: ops.awq_gemm_packed(x, packs[0]))] * 16, iters=40) for _ in range(2))
        two = min(time_graph([(lambda p=p: ops.awq_gemm_packed(x, p)) for p in (packs[0], packs[1]) * 8], iters=40) for _ in range(2))
        print(f"{K}x{N} M={M}: rotating 16 sets {rot:.2f} us, one set {same:.2f} us, two sets alternating {two:.2f} us", flush=True)
    del packs
