"""Developer aid: where a launch shape's output differs from the default's."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import make_weights, make_x
from vllm_quick_amd._C import ops
dev = torch.device("cuda:0")
K, N, G = 4096, 11008, 128
p = ops.awq_repack(*make_weights(K, N, G, 0, dev))
for M in (1280, 2048, 4096):
    x = make_x(M, K, 1, dev)
    ref = ops.awq_gemm_packed(x, p, cfg=("wide", 4, 2, -8)).float()
    for rep in range(25):
        y = ops.awq_gemm_packed(x, p, cfg=("wide", 4, 2, -4)).float()
        bad = (y - ref).abs() > 0.02
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        if int(bad.sum()) or rep == 24: print(M, rep, "bad elems", int(bad.sum()), "row blocks", sorted(set((rows // 128).tolist()))[:20],
              "col slices(256)", sorted(set((cols // 256).tolist()))[:20], flush=True)
