import sys, torch
sys.path.insert(0, "/root/repo")
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
for K, N in ((4096, 5504), (4096, 2752), (4096, 1376), (2048, 11008), (1024, 11008), (512, 11008)):
    packs = [ops.awq_repack(*make_weights(K, N, 128, i, dev)) for i in range(16)]
    x = make_x(128, K, 1, dev)
    us = min(time_graph([(lambda p=p: ops.awq_gemm_packed(x, p)) for p in packs], iters=20) for _ in range(3))
    print(f"{K}x{N} M=128: {us:.2f} us", flush=True)
