"""4-wave decode shapes and depths on very wide layers, single and batched (developer script)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from vllm_quick_amd import _lib as L
from scripts._timing import time_graph
from bench import make_weights, make_x
dev = torch.device("cuda:0")


def launch(x, p, K, N, t, waves, depth, out, epi, batch):
    M = x.shape[-2]
    cols = N // 2 if epi else N
    return L.lib.w4a16_gemm_packed_f16_decode_cfg(x.data_ptr(), 0, p.packed_w.data_ptr(), p.packed_sz.data_ptr(), None,
                                                  out.data_ptr(), M * cols, M, K, N, 128, batch, epi, t, waves, depth, 1,
                                                  None, 0, torch.cuda.current_stream().cuda_stream)


for K, N, epi, batch in ((8192, 57344, 1, 1), (4096, 28672, 0, 1), (5120, 27648, 0, 1), (8192, 28672, 0, 1), (4096, 28672, 1, 8), (14336, 4096, 0, 8),
                         (4096, 28672, 1, 2), (14336, 4096, 0, 2)):
    nb = 2 if batch > 1 else 4
    if batch == 1:
        packs = [ops.awq_repack(*make_weights(K, N, 128, i, dev)) for i in range(nb)]
    else:
        packs = []
        for i in range(nb):
            ws = [make_weights(K, N, 128, i * batch + e, dev) for e in range(batch)]
            packs.append(ops.awq_repack_batch([w[0] for w in ws], [w[1] for w in ws], [w[2] for w in ws]))
    for M in (1, 4, 16):
        x = make_x(M, K, 1, dev)
        out = torch.empty((batch, M, N // 2 if epi else N), dtype=torch.float16, device=dev)
        fn = (ops.awq_gemm_packed_silu_and_mul if epi else (ops.awq_gemm_packed_batched if batch > 1 else ops.awq_gemm_packed))
        us0 = min(time_graph([(lambda p=p: fn(x, p)) for p in packs], iters=10) for _ in range(3))
        res = []
        for t, w in ((2, 4), (4, 4), (4, 8), (2, 8), (8, 8), (6, 8)):
            for d in (1, 2):
                if launch(x, packs[0], K, N, t, w, d, out, epi, batch) != 0:
                    continue
                us = min(time_graph([(lambda p=p: launch(x, p, K, N, t, w, d, out, epi, batch)) for p in packs], iters=10) for _ in range(3))
                res.append((us, (t, w, d)))
        res.sort(key=lambda r: r[0])
        print(f"{K}x{N} epi={epi} batch={batch} M={M}: default {us0:.2f} {L.launch_shape(M, K, N, batch, epi)} | " +
              "  ".join(f"{c}:{u:.2f}" for u, c in res[:7]), flush=True)
