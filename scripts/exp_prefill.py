"""Where does dequantise-to-HBM + fp16 matmul overtake the fused W4A16 GEMM?  (developer script)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
shapes = [(4096, 11008), (4096, 12288), (11008, 4096), (4096, 22016)]
Ms = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "256,512,1024,2048,4096").split(",")]
G = 128
for K, N in shapes:
    ws = [make_weights(K, N, G, i, dev) for i in range(8)]
    packs = [ops.awq_repack(*w) for w in ws]
    for M in Ms:
        x = make_x(M, K, 1, dev)
        fused = time_graph([(lambda p=p: ops.awq_gemm_packed(x, p)) for p in packs])
        def deq_mm(w):
            return torch.matmul(x, ops.awq_dequantize(w[0], w[1], w[2], 0, 0, 0))
        def deqp_mm(p):
            return torch.matmul(x, ops.awq_dequantize_packed(p))
        two = time_graph([(lambda w=w: deq_mm(w)) for w in ws])
        twop = time_graph([(lambda p=p: deqp_mm(p)) for p in packs])
        fl = 2.0 * M * K * N
        print(f"K={K} N={N} M={M}: fused {fused:8.1f} us ({fl/fused*1e-6:6.0f} TF)  dequant+matmul {two:8.1f} us ({fl/two*1e-6:6.0f} TF)  "
              f"packed-dequant+matmul {twop:8.1f} us", flush=True)
