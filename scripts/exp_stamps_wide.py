"""Diagnostic: run the 32x32x16 kernel from a build with -DW4A16_STAMPS and summarise its per-wave cycle
stamps.  Usage: W4A16_LIBRARY=scripts/libw4a16_stamps.so python scripts/exp_stamps_wide.py "2,3,8" 128 [K N]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd import _lib
from vllm_quick_amd._C import ops
from bench import make_weights, make_x

dev = torch.device("cuda:0")
mt, t, kw, *rest = (int(v) for v in sys.argv[1].split(","))
splits = rest[0] if rest else 1
M = int(sys.argv[2]) if len(sys.argv) > 2 else 128
K, N, G = 4096, 11008, 128
if len(sys.argv) > 4:
    K, N = int(sys.argv[3]), int(sys.argv[4])
packs = [ops.awq_repack(*make_weights(K, N, G, i, dev)) for i in range(16)]
x = make_x(M, K, 1, dev)
ws = ops.gemm_workspace(dev, 32 << 20)
for rep in range(40):   # warm clocks, rotate weights so HBM is really read
    for p in packs:
        ops.awq_gemm_packed(x, p, cfg=("wide", mt, t, kw, splits), workspace=ws)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
slices = -(-(-(-N // 32)) // t)
mblocks = -(-M // (mt * 32))
ctas = ((slices * mblocks + 7) // 8) * 8
if splits > 1:
    print("(K split: stamps of the workgroups of split 0 and 1 overwrite each other per blockIdx.x; whichever wrote last is shown)")
buf = np.zeros(4096 * 32, dtype=np.uint64)
rc = lib.w4a16_dbg_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
st = buf.reshape(-1, 32)[: min(ctas * kw, 4096)].astype(np.int64)
wid = np.arange(len(st)) % kw
keep = st[:, 20] > 0
full = st[:, 23] > 0
st, wid, full = st[keep], wid[keep], full[keep]
t0 = st[:, 0].min()
def stat(name, v):
    print(f"{name:34s} min {v.min():8.0f}  med {np.median(v):8.0f}  max {v.max():8.0f}")
print(f"wide ({mt},{t},{kw}) M={M} {K}x{N}: {len(st)} waves stamped")
clk = ((st[:, 23] - st[:, 0]) / np.maximum(st[:, 31] - st[:, 30], 1) * 100.0)[full]
stat("clock MHz (cycles / 100 MHz ticks)", clk)
stat("start skew", st[:, 0] - t0)
stat("setup + first requests", st[:, 1] - st[:, 0])
stat("  entry -> work item known", st[:, 24] - st[:, 0])
stat("  -> offsets, accumulators zeroed", st[:, 25] - st[:, 24])
stat("  -> first weight block requested", st[:, 26] - st[:, 25])
stat("  -> half slab 0 requested", st[:, 27] - st[:, 26])
stat("  -> half slab 1 requested", st[:, 1] - st[:, 27])
nblk = -(-(-(-(-(-K // 128)) // splits)) // kw) + (1 if kw == 8 else 0)
prev = st[:, 1]
for b in range(min(nblk, 16)):
    cur = st[:, 2 + b]
    ok = cur > 0
    stat(f"block {b}", (cur - prev)[ok])
    prev = cur
stat("main loop", st[:, 20] - st[:, 1])
stat("wait at first barrier", st[:, 21] - st[:, 20])
stat("fold / park partials", (st[:, 22] - st[:, 21])[full])
stat("sum + store", (st[:, 23] - st[:, 22])[full])
stat("wave lifetime", (st[:, 23] - st[:, 0])[full])
stat("entry -> main loop done", st[:, 20] - st[:, 0])
for wv in range(kw):
    sel = st[wid == wv]
    print(f" wave {wv}: first-req {np.median(sel[:, 1] - sel[:, 0]):6.0f} | blocks " +
          " ".join(f"{np.median(sel[:, 2 + b] - (sel[:, 1 + b] if b else sel[:, 1])):6.0f}" for b in range(min(nblk, 16))) +
          f" | loop-done@ {np.median(sel[:, 20] - sel[:, 0]):6.0f} barrier {np.median(sel[:, 21] - sel[:, 20]):6.0f} fold {np.median(sel[:, 22] - sel[:, 21]):6.0f} store {np.median(sel[:, 23] - sel[:, 22]):6.0f}")
