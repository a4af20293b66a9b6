import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
cfgs = [tuple(int(v) for v in c.split(",")) for c in sys.argv[1].split(";")]
Ms = [int(v) for v in sys.argv[2].split(",")]
K, N, G = 4096, 11008, 128
packs = []
for i in range(16):
    qw, s, qz = make_weights(K, N, G, i, dev)
    packs.append(ops.awq_repack(qw, s, qz))
for M in Ms:
    x = make_x(M, K, 1, dev)
    for cfg in cfgs:
        us = time_graph([(lambda p=p: ops.awq_gemm_packed(x, p, cfg=cfg)) for p in packs])
        print(f"{os.environ.get('W4A16_LIBRARY','default')[-12:]} M={M} cfg={cfg}: {us:.2f} us  {2.0*M*K*N/us*1e-6:.1f} TFLOP/s", flush=True)
