"""Diagnostic: run the shared-slab 32x32x16 kernel from a build with -DW4A16_STAMPS on w4a16_wide_shared.hip and
summarise its per-wave cycle stamps (non-staggered shapes).
Usage: W4A16_LIBRARY=scripts/_exp/libstamps_shared.so python scripts/exp_stamps_shared.py "4,3,4" 1024 [K N]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd import _lib
from vllm_quick_amd._C import ops
from bench import make_weights, make_x

dev = torch.device("cuda:0")
mt, t, nw = (int(v) for v in sys.argv[1].split(","))
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
K, N, G = 4096, 11008, 128
if len(sys.argv) > 4:
    K, N = int(sys.argv[3]), int(sys.argv[4])
packs = [ops.awq_repack(*make_weights(K, N, G, i, dev)) for i in range(4)]
x = make_x(M, K, 1, dev)
for rep in range(10):
    for p in packs:
        ops.awq_gemm_packed(x, p, cfg=("wide", mt, t, -nw))
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(4096 * 32, dtype=np.uint64)
rc = lib.w4a16_dbg_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
st = buf.reshape(-1, 32).astype(np.int64)
st = st[(st[:, 27] > 0) & (st[:, 0] > 0)]
def stat(name, v):
    print(f"{name:40s} min {v.min():8.0f}  med {np.median(v):8.0f}  max {v.max():8.0f}")
print(f"shared ({mt},{t},{nw}) M={M} {K}x{N}: {len(st)} waves stamped")
stat("clock MHz", (st[:, 27] - st[:, 0]) / np.maximum(st[:, 31] - st[:, 30], 1) * 100.0)
stat("entry -> first requests out", st[:, 1] - st[:, 0])
for b in range(8):
    a0, a1, a2 = st[:, 2 + 3 * b], st[:, 3 + 3 * b], st[:, 4 + 3 * b]
    nxt = st[:, 2 + 3 * (b + 1)] if b < 7 else None
    line = f"block {b}: wait-own-DMA {np.median(a1 - a0):6.0f}  barrier {np.median(a2 - a1):6.0f}"
    if nxt is not None:
        line += f"  compute {np.median(nxt - a2):6.0f}   (block total {np.median(nxt - a0):6.0f})"
    print(line)
stat("main loop (first block -> loop end)", st[:, 26] - st[:, 2])
stat("stores", st[:, 27] - st[:, 26])
stat("wave lifetime", st[:, 27] - st[:, 0])
