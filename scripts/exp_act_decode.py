"""Fused gate_up + activation at 1-16 rows: every decode shape and depth against the default launch (developer script)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from vllm_quick_amd import _lib as L
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(4096, 22016), (4096, 28672), (5120, 27648), (8192, 7168)]


def launch(x, p, K, N, t, waves, depth, out):
    M = x.shape[0]
    rc = L.lib.w4a16_gemm_packed_f16_decode_cfg(x.data_ptr(), 0, p.packed_w.data_ptr(), p.packed_sz.data_ptr(), None,
                                                out.data_ptr(), M * (N // 2), M, K, N, 128, 1, 1, t, waves, depth, 1,
                                                None, 0, torch.cuda.current_stream().cuda_stream)
    return rc


for K, N in shapes:
    nb = max(4, min(16, int(400e6 / (K * N / 2))))
    packs = [ops.awq_repack(*make_weights(K, N, 128, i, dev)) for i in range(nb)]
    for M in (1, 4, 8, 16):
        x = make_x(M, K, 1, dev)
        ref = ops.awq_gemm_packed_silu_and_mul(x, packs[0]).float()
        out = torch.empty((M, N // 2), dtype=torch.float16, device=dev)
        us0 = min(time_graph([(lambda p=p: ops.awq_gemm_packed_silu_and_mul(x, p)) for p in packs], iters=20) for _ in range(3))
        res = []
        for t, w in ((2, 16), (2, 8), (4, 8), (6, 8), (8, 8), (2, 4), (4, 4)):
            for d in (1, 2):
                if launch(x, packs[0], K, N, t, w, d, out) != 0:
                    continue
                err = (out.float() - ref).abs().max().item()
                us = min(time_graph([(lambda p=p: launch(x, p, K, N, t, w, d, out)) for p in packs], iters=20) for _ in range(3))
                res.append((us, (t, w, d), err))
        res.sort(key=lambda r: r[0])
        print(f"{K}x{N} M={M}: default {us0:.2f} {L.launch_shape(M, K, N, 1, 1)} | " +
              "  ".join(f"{c}:{u:.2f}({e:.0e})" for u, c, e in res[:6]), flush=True)
