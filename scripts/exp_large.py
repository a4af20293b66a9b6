"""Shared-slab 32x32x16 shapes at many rows: checks each against a dense fp32 product, then times them in turn."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd import _lib
from vllm_quick_amd._C import ops
from bench import make_weights, make_x
from scripts._timing import time_graph

dev = torch.device("cuda:0")
K, N, G = 4096, 11008, 128
stock = [make_weights(K, N, G, i, dev) for i in range(4)]
packs = [ops.awq_repack(*w) for w in stock]
dense = ops.awq_dequantize(*stock[0], 0, 0, 0).float()
tag = os.path.basename(os.environ.get("W4A16_LIBRARY", "default"))
for M in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "512,1024,4096").split(",")]:
    x = make_x(M, K, 1, dev)
    want = x.float() @ dense
    cfgs = [None] + [("wide", mt, t, -nw) for mt, t, nw in _lib.built_shapes(3)]
    times = {str(c): [] for c in cfgs}
    for c in cfgs:
        y = ops.awq_gemm_packed(x, packs[0], cfg=c)
        err = (y.float() - want).abs().max().item()
        assert err < 0.05, (c, err)
        assert all(torch.equal(ops.awq_gemm_packed(x, packs[0], cfg=c), y) for _ in range(3)), c
    for rnd in range(4):
        for c in cfgs:
            times[str(c)].append(time_graph([(lambda p=p, c=c: ops.awq_gemm_packed(x, p, cfg=c)) for p in packs], iters=6))
    for c in cfgs:
        t = sorted(times[str(c)][1:])
        print(f"{tag} M={M} {str(c):22s} median {t[len(t) // 2]:8.2f} us  {2.0 * M * K * N / t[len(t) // 2] * 1e-6:7.1f} TFLOP/s", flush=True)
