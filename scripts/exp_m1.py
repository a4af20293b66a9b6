"""Decode kernel at a few rows on one layer: every built (tiles, waves) shape at both prefetch depths, timed in turn
over several rounds (graph replays over 16 rotating weight sets)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd import _lib
from vllm_quick_amd._C import ops
from bench import make_weights, make_x
from scripts._timing import time_graph

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K, N, G = (int(sys.argv[2]), int(sys.argv[3]), 128) if len(sys.argv) > 3 else (4096, 11008, 128)
packs = [ops.awq_repack(*make_weights(K, N, G, i, dev)) for i in range(16)]
x = make_x(M, K, 1, dev)
cfgs = [None] + [("decode", t, w, 1, d) for t, w, _ in _lib.built_shapes(4) for d in (1, 2)]
ok = []
for c in cfgs:
    try:
        ops.awq_gemm_packed(x, packs[0], cfg=c)
        ok.append(c)
    except Exception as e:  # noqa: BLE001
        print(c, "->", type(e).__name__, flush=True)
times = {str(c): [] for c in ok}
for rnd in range(4):
    for c in ok:
        times[str(c)].append(time_graph([(lambda p=p, c=c: ops.awq_gemm_packed(x, p, cfg=c)) for p in packs], iters=10))
for c in ok:
    t = sorted(times[str(c)][1:])
    print(f"{str(c):28s} min {t[0]:6.2f}  median {t[len(t) // 2]:6.2f} us", flush=True)
