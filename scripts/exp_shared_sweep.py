"""Developer aid: every 32x32x16 launch shape (private slabs and shared slab) on several layer shapes and row
counts (JSON lines), the data the launch-shape choosers' constants are fitted on."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd import _lib
from vllm_quick_amd._C import ops
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
G = 128
shapes = [("wide", ) + c for c in _lib.built_shapes(2)] + [("wide", mt, t, -nw) for mt, t, nw in _lib.built_shapes(3)]
layers = [(4096, 11008), (4096, 12288), (4096, 4096), (11008, 4096), (4096, 22016), (8192, 7168), (8192, 1280),
          (1024, 8192), (3584, 8192), (4096, 14336), (14336, 4096), (4096, 1376), (4096, 5504), (2048, 11008)]
rows = [int(a) for a in sys.argv[1:]] or [160, 256, 320, 512, 768, 1024, 1536, 2048, 4096]
for K, N in layers:
    # weight sets of at least 640 MB in all, rotated: beyond the 256 MB memory-side cache, as a model's layers are
    # (four distinct sets, the rest are copies in memory of their own)
    n = max(16, -(-(640 << 20) // (K * N // 2)))
    base = [ops.awq_repack(*make_weights(K, N, G, i, dev)) for i in range(4)]
    packs = []
    for i in range(n):
        b = base[i % 4]
        p = type(b).__new__(type(b))
        p.__dict__.update(b.__dict__)
        p.packed_w, p.packed_sz = b.packed_w.clone(), b.packed_sz.clone()
        packs.append(p)
    for M in rows:
        x = make_x(M, K, 1, dev)
        res = {}
        for cfg in [None] + shapes:
            it = 2 if M <= 1024 else 1
            us = min(time_graph([(lambda p=p: ops.awq_gemm_packed(x, p, cfg=cfg)) for p in packs], iters=it) for _ in range(2))
            res[str(cfg)] = us
        print("JSON " + json.dumps({"K": K, "N": N, "M": M, "default": _lib.launch_shape(M, K, N), "us": res}), flush=True)
    del packs, base
    torch.cuda.empty_cache()
