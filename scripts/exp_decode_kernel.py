"""Decode kernel against the streaming kernel on one layer shape: correctness (bit-compare against the streaming
kernel at the same launch shape is NOT expected: the sum order over k differs only across waves, which is the
same split, so results should be identical) and graph-replayed times.  Usage: exp_decode_kernel.py K N [M..]"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from vllm_quick_amd import _lib
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
K, N = int(sys.argv[1]), int(sys.argv[2])
Ms = [int(v) for v in sys.argv[3:]] or [1, 2, 4]
nb = max(4, min(16, int(400e6 / (K * N / 2))))
packs = [ops.awq_repack(*make_weights(K, N, 128, i, dev)) for i in range(nb)]
ws = ops.gemm_workspace(dev)
for M in Ms:
    x = make_x(M, K, 1, dev)
    cands = [("old", (1, t, w)) for t, w in ((1, 16), (2, 16), (3, 16), (2, 8), (3, 8), (4, 8))]
    cands += [("dec", ("decode", t, w, s, d)) for t, w in ((1, 16), (2, 16), (3, 16), (1, 8), (2, 8), (3, 8), (4, 8)) for s in (1, 2, 3) for d in (1, 2)]
    ref = ops.awq_gemm_packed(x, packs[0], cfg=(1, 1, 16)).float()
    res = []
    for name, cfg in cands:
        try:
            y = ops.awq_gemm_packed(x, packs[0], cfg=cfg, workspace=ws if name == "dec" else None).float()
        except Exception as e:
            continue
        err = (y - ref).abs().max().item()
        us = min(time_graph([(lambda p=p: ops.awq_gemm_packed(x, p, cfg=cfg, workspace=ws if name == "dec" else None)) for p in packs], iters=20) for _ in range(2))
        res.append((us, name, cfg, err))
    res.sort(key=lambda r: r[0])
    print(f"{K}x{N} M={M}: " + "  ".join(f"{c}:{u:.2f}({e:.1e})" for u, n, c, e in res[:8]), flush=True)
    print("   old best: " + "  ".join(f"{c}:{u:.2f}" for u, n, c, e in res if n == "old"), flush=True)
    print("   all dec: " + "  ".join(f"{c[1:]}:{u:.2f}" for u, n, c, e in sorted(res, key=lambda r: r[2][1:] if r[1] == "dec" else (0,)) if n == "dec"), flush=True)
    print("   bad: " + str([(c, e) for u, n, c, e in res if e > 2e-3]), flush=True)
