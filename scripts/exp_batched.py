"""Launch-shape sweep of the batched W4A16 GEMM (developer script): steady-state weight bandwidth."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
K, N, G = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "4096,14336,128").split(","))
E = int(sys.argv[2]) if len(sys.argv) > 2 else 16
Ms = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "1,8").split(",")]
cfgs = [None] + [tuple(int(v) for v in c.split(",")) for c in (sys.argv[4] if len(sys.argv) > 4 else "1,3,16;1,3,8;1,4,8;1,6,8;1,8,8;1,2,8;1,4,4").split(";")]
sets = []
for r in range(2):
    ws = [make_weights(K, N, G, r * E + i, dev) for i in range(E)]
    sets.append(ops.awq_repack_batch([w[0] for w in ws], [w[1] for w in ws], [w[2] for w in ws]))
    del ws
wbytes = E * (K * N // 2 + (K // G) * N * 2 + (K // G) * N // 2)
for M in Ms:
    x = make_x(M, K, 1, dev)
    for cfg in cfgs:
        if cfg is not None and cfg[0] != -(-M // 16) and len(cfg) == 3:
            cfg = (-(-M // 16),) + cfg[1:]
        try:
            us = time_graph([(lambda p=p: ops.awq_gemm_packed_batched(x, p, cfg=cfg)) for p in sets])
            print(f"E={E} M={M} cfg={cfg}: {us:8.1f} us  {wbytes/us*1e-6:6.2f} TB/s weights", flush=True)
        except Exception as e:
            print(f"E={E} M={M} cfg={cfg}: {type(e).__name__}: {e}", flush=True)
