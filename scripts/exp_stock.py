"""Developer aid: time of the stock-layout GEMM kernel against the packed route, 4096 x 11008 g128."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_weights, make_x, time_launches
from vllm_quick_amd._C import ops

dev = torch.device("cuda:0")
K, N, G = 4096, 11008, 128
stock = [make_weights(K, N, G, seed=i, device=dev) for i in range(8)]
packed = [ops.awq_repack(*w) for w in stock]
stream = torch.cuda.Stream(device=dev)
for m in (1, 8, 16, 32, 64, 128, 512):
    x = make_x(m, K, seed=m, device=dev)
    a = time_launches([(lambda w=w: ops.awq_gemm_stock(x, *w)) for w in stock], stream, 64, 8)
    b = time_launches([(lambda p=p: ops.awq_gemm_packed(x, p)) for p in packed], stream, 64, 8)
    print(f"M={m:4d} stock {a:8.1f} us   packed {b:8.1f} us", flush=True)
