"""Default-launch kernel times at decode row counts on the Llama-2-7B layer shapes and the benchmark layer,
with a check against the stock-layout kernel (developer script).  usage: exp_decode.py [rows ...]"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
rows = [int(a) for a in sys.argv[1:]] or [1, 4, 8, 16]
out = {}
for (K, N) in ((4096, 11008), (4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)):
    ws = [make_weights(K, N, 128, i, dev) for i in range(8 if K * N > 60e6 else 16)]
    packs = [ops.awq_repack(*w) for w in ws]
    line = []
    for M in rows:
        x = make_x(M, K, 1, dev)
        err = (ops.awq_gemm_packed(x, packs[0]).float() - ops.awq_gemm_stock(x, *ws[0]).float()).abs().max().item()
        us = min(time_graph([(lambda p=p: ops.awq_gemm_packed(x, p)) for p in packs], iters=20) for _ in range(3))
        line.append(f"M={M}: {us:.2f}us(d{err:.0e})")
        out[f"{K}x{N}/{M}"] = us
    print(f"{K}x{N}: " + "  ".join(line), flush=True)
print("JSON " + json.dumps(out))
