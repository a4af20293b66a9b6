"""Diagnostic: run the tiled kernel from a build with -DW4A16_STAMPS and summarise its per-wave
cycle stamps.  Usage: W4A16_LIBRARY=scripts/libw4a16_stamps.so python scripts/exp_stamps.py "4,3,2,4" 128"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd import _lib
from vllm_quick_amd._C import ops
from bench import make_weights, make_x

dev = torch.device("cuda:0")
cfg = tuple(int(v) for v in sys.argv[1].split(","))
M = int(sys.argv[2]) if len(sys.argv) > 2 else 128
K, N, G = 4096, 11008, 128
packs = [ops.awq_repack(*make_weights(K, N, G, i, dev)) for i in range(16)]
x = make_x(M, K, 1, dev)
for rep in range(40):   # warm clocks, rotate weights so HBM is really read
    for p in packs:
        ops.awq_gemm_packed(x, p, cfg=cfg)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
mt, t, nw, kw = cfg
nwaves = nw * kw
slices = -(-(-(-N // 16)) // (nw * t))
mblocks = -(-M // (mt * 16))
ctas = ((slices + 7) // 8) * 8 * mblocks
buf = np.zeros(4096 * 32, dtype=np.uint64)
rc = lib.w4a16_dbg_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
st = buf.reshape(-1, 32)[: ctas * nwaves].astype(np.int64)
wid = np.arange(len(st)) % nwaves
keep = st[:, 1] > 0
st, wid = st[keep], wid[keep]
t0 = st[:, 0].min()
def stat(name, v):
    print(f"{name:28s} min {v.min():8.0f}  med {np.median(v):8.0f}  max {v.max():8.0f}")
print(f"cfg {cfg} M={M}: {len(st)} waves stamped")
stat("start skew (cycles)", st[:, 0] - t0)
stat("prologue", st[:, 1] - st[:, 0])
stat("loop", st[:, 2] - st[:, 1])
stat("  work in phases", st[:, 5])
stat("  barrier wait", st[:, 6])
stat("reduce write+barrier", st[:, 3] - st[:, 2])
stat("epilogue", st[:, 4] - st[:, 3])
stat("wave lifetime", st[:, 4] - st[:, 0])
print(f"first start -> last end: {st[:, 4].max() - t0} cycles")

iters = min(8, -(-(-(-K // 512)) * 4) // kw)
print("per wave index: median work / barrier per phase")
for wv in range(nwaves):
    sel = st[wid == wv]
    print(f" w{wv} (kg {wv // nw} nw {wv % nw}): total work {np.median(sel[:, 5]):7.0f} bar {np.median(sel[:, 6]):7.0f} | " +
          " ".join(f"{np.median(sel[:, 8 + 2 * i]):5.0f}/{np.median(sel[:, 9 + 2 * i]):4.0f}" for i in range(iters)))
one = st[:nwaves]
print("first workgroup, per wave per phase work/barrier:")
for wv in range(len(one)):
    print(f" w{wv}: " + " ".join(f"{one[wv, 8 + 2 * i]:5d}/{one[wv, 9 + 2 * i]:4d}" for i in range(iters)))

stat("prologue: setup", st[:, 24] - st[:, 0])
stat("prologue: loads issued", st[:, 25] - st[:, 24])
stat("prologue: x landed", st[:, 26] - st[:, 25])
stat("prologue: stored+barrier", st[:, 1] - st[:, 26])
