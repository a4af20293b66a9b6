"""The shard shapes bench.py launches at 2/4/8 ranks (column shards [4096, 11008/p], row shards
[4096/p, 11008], M = 128 and a few decode sizes) against the oracle.  Developer check, GPU box."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import awq_oracle as orc
from vllm_quick_amd._C import ops
dev = torch.device("cuda:0")
bad = 0
for p in (2, 4, 8):
    for (K, N) in ((4096, 11008 // p), (4096 // p, 11008)):
        for M in (1, 16, 32, 128):
            x, qw, s, qz = orc.make_inputs(M, K, N, 128, seed=p + M)
            want = orc.gemm_faithful(x, qw, s, qz).astype(np.float32)
            exact = orc.gemm_exact(x, qw, s, qz)
            tx, tq, ts, tz = [torch.from_numpy(a).to(dev) for a in (x, qw, s, qz)]
            y = ops.awq_gemm(tx, tq, ts, tz, 8).float().cpu().numpy()
            ok = np.isclose(y, want, rtol=1e-2, atol=1e-3)
            frac = 1 - ok.mean()
            # the reference arithmetic's own fp16 slice noise can push a near-zero output or two outside the
            # bound (see tests/test_gemm_gpu.py: assert_parity); those must be within it of the fp64 result
            bad += (~ok).sum() > 2 or not np.isclose(y[~ok], exact[~ok], rtol=1e-2, atol=1e-3).all()
            print(f"p={p} K={K} N={N} M={M}: outside tolerance {frac:.2e} max|d| {np.abs(y - want).max():.2e}", flush=True)
print("FAIL" if bad else "ALL OK")
sys.exit(1 if bad else 0)
