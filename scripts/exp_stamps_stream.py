"""Diagnostic: run the streaming kernel from a build with -DW4A16_STAMPS and summarise its per-wave
cycle stamps.  Usage: W4A16_LIBRARY=scripts/_libs/libw4a16_stamps.so python scripts/exp_stamps_stream.py "1,3,16" 1"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd import _lib
from vllm_quick_amd._C import ops
from bench import make_weights, make_x

dev = torch.device("cuda:0")
decode = sys.argv[1].startswith("d")   # "d3,16": the decode kernel with t = 3, 16 waves
cfg = tuple(int(v) for v in sys.argv[1].lstrip("d").split(","))
if decode:
    cfg = ("decode", cfg[0], cfg[1], 1)
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1
K, N, G = 4096, 11008, 128
packs = [ops.awq_repack(*make_weights(K, N, G, i, dev)) for i in range(16)]
x = make_x(M, K, 1, dev)
for rep in range(40):   # warm clocks, rotate weights so HBM is really read
    for p in packs:
        ops.awq_gemm_packed(x, p, cfg=cfg)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
mt, t, W = (1, cfg[1], cfg[2]) if decode else cfg
ctas = -(-(-(-N // 16)) // t) * -(-M // (mt * 16))
buf = np.zeros(4096 * 32, dtype=np.uint64)
rc = lib.w4a16_dbg_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
st = buf.reshape(-1, 32)[: ctas * W].astype(np.int64)
wid = np.arange(len(st)) % W
blocks = -(-K // 128) // W
# the counter is per XCD: every workgroup is measured from the start of its own first wave
t0 = np.repeat(st[:, 0].reshape(-1, W).min(axis=1), W)
def stat(name, v):
    print(f"{name:34s} min {v.min():8.0f}  med {np.median(v):8.0f}  p90 {np.percentile(v, 90):8.0f}  max {v.max():8.0f}")
print(f"cfg {cfg} M={M}: {len(st)} waves, {blocks} blocks per wave")
stat("start after first wave", st[:, 0] - t0)
stat("setup + loads issued", st[:, 1] - st[:, 0])
if decode:
    stat("first block's weights requested", st[:, 13] - st[:, 0])
    stat("address setup done", st[:, 14] - st[:, 0])
    stat("LDS-DMA issued", st[:, 15] - st[:, 0])
    for wv in (0, 4, 8, 12):
        sel = st[wid == wv]
        print(f" w{wv}: first weights {np.median(sel[:, 13] - sel[:, 0]):.0f} setup {np.median(sel[:, 14] - sel[:, 0]):.0f} dma {np.median(sel[:, 15] - sel[:, 0]):.0f} all requested {np.median(sel[:, 1] - sel[:, 0]):.0f}")
prev = st[:, 1]
for b in range(min(blocks, 8)):
    stat(f"block {b} done after previous", st[:, 2 + b] - prev)
    stat(f"block {b} done since first wave", st[:, 2 + b] - t0)
    prev = st[:, 2 + b]
stat("loop end since first wave", st[:, 10] - t0)
stat("LDS write + barrier", st[:, 11] - st[:, 10])
stat("sum + store", st[:, 12] - st[:, 11])
stat("wave lifetime", st[:, 12] - st[:, 0])
stat("end since first wave", st[:, 12] - t0)
print("per wave index: median 'block done since first wave' / loop end")
for wv in range(W):
    sel = st[wid == wv]
    print(f" w{wv:2d}: start {np.median(sel[:, 0] - t0[wid == wv]):6.0f} issued {np.median(sel[:, 1] - t0[wid == wv]):6.0f} | " + " ".join(f"{np.median(sel[:, 2 + b] - t0[wid == wv]):7.0f}" for b in range(min(blocks, 8))) +
          f" | loop end {np.median(sel[:, 10] - t0[wid == wv]):7.0f} barrier {np.median(sel[:, 11] - t0[wid == wv]):7.0f} end {np.median(sel[:, 12] - t0[wid == wv]):7.0f}")
