import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
from oracle import awq_oracle as orc
dev = torch.device("cuda:0")
cfgs = [tuple(int(v) for v in c.split(",")) for c in sys.argv[1].split(";")]
Ms = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [128]
# correctness on a ragged case vs the streaming kernel and the oracle
for (M, K, N, G) in [(70, 640, 400, 128), (128, 1024, 1376, 64), (130, 352, 96, 32)]:
    x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=7)
    want = orc.gemm_faithful(x, qw, s, qz).astype(np.float32)
    tx, tqw, ts, tqz = [torch.from_numpy(a).to(dev) for a in (x, qw, s, qz)]
    pk = ops.awq_repack(tqw, ts, tqz)
    for cfg in cfgs:
        try:
            y = ops.awq_gemm_packed(tx, pk, cfg=cfg).float().cpu().numpy()
            bad = ~np.isclose(y, want, rtol=1e-2, atol=1e-3)
            print(f"check {M}x{K}x{N} g{G} cfg={cfg}: mismatches {bad.sum()} maxdiff {np.abs(y-want).max():.3e}", flush=True)
        except Exception as e:
            print(f"check cfg={cfg}: {type(e).__name__}: {e}", flush=True)
K, N, G = 4096, 11008, 128
packs = []
for i in range(16):
    qw, s, qz = make_weights(K, N, G, i, dev)
    packs.append(ops.awq_repack(qw, s, qz))
for M in Ms:
    x = make_x(M, K, 1, dev)
    for cfg in cfgs:
        try:
            us = time_graph([(lambda p=p: ops.awq_gemm_packed(x, p, cfg=cfg)) for p in packs])
            print(f"M={M} cfg={cfg}: {us:.2f} us  {2.0*M*K*N/us*1e-6:.1f} TFLOP/s", flush=True)
        except Exception as e:
            print(f"M={M} cfg={cfg}: {type(e).__name__}: {e}", flush=True)
